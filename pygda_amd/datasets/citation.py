"""``CitationDataset`` file-format reader (pygda/datasets/citation.py:110-204): the
ACMv9 / Citationv1 / DBLPv7 release as three text files under ``<root>/raw``:

  ``{name}_edgelist.txt``  one ``a,b`` integer pair per line  -> edge_index [2, E] int64
  ``{name}_docs.txt``      one comma-separated float row per node -> x [N, F] float32
  ``{name}_labels.txt``    one integer per line                 -> y [N] int64

plus random 80/10/10 train/val/test masks (unseeded ``np.random.permutation`` in the
reference, :178-194; unused by the trainers on the hot path).  The parsed tensors are cached
as one binary file under ``<root>/processed`` so the text is read once.
"""
import os
import os.path as osp

import numpy as np
import torch

from ..data import Data


def _read_matrix(path, dtype):
    rows = []
    with open(path, "rb") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append(np.array(line.split(b","), dtype=dtype))   # whole-row C conversion
    return np.vstack(rows) if rows else np.zeros((0, 0), dtype=dtype)


class CitationDataset:
    def __init__(self, root, name, transform=None, pre_transform=None, pre_filter=None):
        self.root, self.name = root, name
        self.transform, self.pre_transform, self.pre_filter = transform, pre_transform, pre_filter
        self.raw_dir, self.processed_dir = osp.join(root, "raw"), osp.join(root, "processed")
        if not osp.exists(self.processed_paths[0]):
            os.makedirs(self.processed_dir, exist_ok=True)
            self.process()
        self.data = Data(**torch.load(self.processed_paths[0]))

    @property
    def raw_file_names(self):
        return [f"{self.name}_docs.txt", f"{self.name}_edgelist.txt", f"{self.name}_labels.txt"]

    @property
    def processed_file_names(self):
        return ["data.pt"]

    @property
    def processed_paths(self):
        return [osp.join(self.processed_dir, f) for f in self.processed_file_names]

    def download(self):
        """The files are distributed out of band (data/README.md of the reference)."""

    def process(self):
        for f in self.raw_file_names:
            if not osp.exists(osp.join(self.raw_dir, f)):
                raise FileNotFoundError(osp.join(self.raw_dir, f))
        edges = _read_matrix(osp.join(self.raw_dir, f"{self.name}_edgelist.txt"), np.int64)
        edge_index = torch.from_numpy(edges.reshape(-1, 2).T.copy())
        x = torch.from_numpy(_read_matrix(osp.join(self.raw_dir, f"{self.name}_docs.txt"), np.float64)
                             ).to(torch.float)
        with open(osp.join(self.raw_dir, f"{self.name}_labels.txt"), "rb") as f:
            y = torch.from_numpy(np.array([l.strip() for l in f if l.strip()], dtype=np.int64))
        n = y.shape[0]
        perm = np.random.permutation(n)
        n_train, n_val = int(n * 0.8), int(n * 0.1)
        masks = {}
        for key, idx in (("train_mask", perm[:n_train]), ("val_mask", perm[n_train:n_train + n_val]),
                         ("test_mask", perm[n_train + n_val:])):
            m = torch.zeros(n, dtype=torch.bool)
            m[idx] = True
            masks[key] = m
        data = Data(edge_index=edge_index, x=x, y=y, **masks)
        if self.pre_transform is not None:
            data = self.pre_transform(data, self.processed_paths[0])
        torch.save({k: data[k] for k in data.keys()}, self.processed_paths[0])

    # -- the bits of the InMemoryDataset protocol the benchmark scripts use --------------
    def __len__(self):
        return 1

    def __getitem__(self, idx):
        if idx != 0:
            raise IndexError(idx)
        return self.data if self.transform is None else self.transform(self.data)

    @property
    def num_classes(self):
        return int(self.data.y.max()) + 1

    @property
    def num_node_features(self):
        return self.data.x.size(1)

    num_features = num_node_features
