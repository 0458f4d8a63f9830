"""Stand-in for dgl.dataloading: GraphDataLoader is a torch DataLoader (the collate function does the batching)."""
from torch.utils.data import DataLoader as GraphDataLoader  # noqa: F401
