from .citation import CitationDataset

__all__ = ["CitationDataset"]
