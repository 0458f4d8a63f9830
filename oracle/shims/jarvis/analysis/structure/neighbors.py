class NeighborsAnalysis:
    pass
