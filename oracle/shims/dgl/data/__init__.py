"""``dgl.data`` stub (test infrastructure only): graphs.py subclasses DGLDataset."""


class DGLDataset:
    def __init__(self, name=None, **_):
        self.name = name
