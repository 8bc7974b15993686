"""``svd_transform`` (pygda/utils/svd_transform.py:8-75): the dataset pre-transform SpecReg relies
on.  Attaches ``data.eival`` (sqrt of the explained variance) and ``data.eivec`` ([k, N] principal
directions) of the dense combinatorial Laplacian ``D - A``; k = 100 below 1000 nodes, else 1000.
Host-side, one-off per dataset, same estimator and seed as the reference (sklearn
``TruncatedSVD(n_iter=20, random_state=42)``); the files ``<processed_paths>eival.pt / eivec.pt``
are written as there."""
import numpy as np
import torch


def laplacian_dense(edge_index, num_nodes):
    """PyG ``get_laplacian(edge_index, normalization=None)`` scattered into a dense matrix the way
    svd_transform.py:59-64 does: self loops dropped, ``-1`` per edge (last writer wins for
    duplicates), degree on the diagonal."""
    row, col = edge_index[0].cpu().numpy(), edge_index[1].cpu().numpy()
    keep = row != col
    row, col = row[keep], col[keep]
    adj = np.zeros((num_nodes, num_nodes), dtype=np.float32)
    adj[row, col] = -1.0
    deg = np.bincount(row, minlength=num_nodes).astype(np.float32)
    adj[np.arange(num_nodes), np.arange(num_nodes)] = deg
    return adj


def svd_transform(data, processed_paths=None):
    from sklearn.decomposition import TruncatedSVD
    num_node = data.y.shape[0]
    adj = laplacian_dense(data.edge_index, num_node)
    pca = TruncatedSVD(n_components=100 if num_node < 1000 else 1000, n_iter=20, random_state=42)
    pca.fit(adj)
    data.eival = torch.tensor(pca.explained_variance_ ** 0.5, dtype=torch.float32)
    data.eivec = torch.tensor(pca.components_, dtype=torch.float32)
    if processed_paths is not None:
        torch.save(data.eival, processed_paths + 'eival.pt')
        torch.save(data.eivec, processed_paths + 'eivec.pt')
    return data
