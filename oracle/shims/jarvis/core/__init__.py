"""Empty ``jarvis`` stub so ``alignn/graphs.py`` imports (test infrastructure only)."""
