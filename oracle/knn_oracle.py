"""CPU restatement of STEGO's KNN precompute (TEST INFRASTRUCTURE ONLY - never imported by the product path).

Follows /root/reference/src/precompute_knns.py:
  :15-21  get_feats: feats = F.normalize(model(img).mean([2, 3]), dim=1)      (F.normalize eps = 1e-12)
  :86-96  n_batches = 16 row blocks; pairwise_sims = einsum("nf,mf->nm", block, normed_feats);
          all_nns.append(torch.topk(pairwise_sims, 30)[1]); cat -> int64 [N, 30]; saved as npz key `nns`.
The top-k itself is torch.topk (ATen; tie order unspecified), restated here with a stable sort: descending
similarity, ties by ascending index.

Parity status: PINNED - tests/golden/knn_small.npz is produced by oracle/make_golden.py running the reference's
own op sequence (torch.einsum + torch.topk, lines 90-91) and this restatement is checked against it in
tests/test_oracle_golden.py.
"""
import numpy as np


def normalize_rows(x, eps=1e-12):
    """F.normalize(x, dim=1) (precompute_knns.py:19)."""
    x = np.asarray(x, dtype=np.float64)
    n = np.sqrt((x * x).sum(axis=1, keepdims=True))
    return x / np.maximum(n, eps)


def knn_topk(x, k=30, normalize=False, q_begin=0, q_count=None, block=4096):
    """Returns (idx int64 [q,k], sims float64 [q,k]) - precompute_knns.py:88-93, any row blocking."""
    x = np.asarray(x, dtype=np.float64)
    if normalize:
        x = normalize_rows(x)
    n = x.shape[0]
    q_count = n - q_begin if q_count is None else q_count
    idx = np.empty((q_count, k), dtype=np.int64)
    val = np.empty((q_count, k), dtype=np.float64)
    for r0 in range(0, q_count, block):
        r1 = min(q_count, r0 + block)
        s = x[q_begin + r0: q_begin + r1] @ x.T                     # einsum("nf,mf->nm")
        order = np.argsort(-s, axis=1, kind="stable")[:, :k]         # topk: descending, ties by index
        idx[r0:r1] = order
        val[r0:r1] = np.take_along_axis(s, order, axis=1)
    return idx, val


def check_neighbours(x, idx, k, sims=None, tol=5e-6, q_begin=0, sim_tol=2e-5):
    """Order-insensitive parity check used by the GPU tests: the returned neighbours must have the same
    similarities (within tol) as the exact top-k, i.e. they may differ from it only inside a tie band.
    Returns the number of rows whose index SET differs from the exact one (informational)."""
    x = np.asarray(x, dtype=np.float64)
    ref_idx, ref_val = knn_topk(x, k, q_begin=q_begin, q_count=idx.shape[0])
    got_val = np.einsum("qkd,qd->qk", x[idx], x[q_begin: q_begin + idx.shape[0]])
    # (a) the similarities are the top-k similarities, in descending order
    np.testing.assert_allclose(got_val, ref_val, rtol=0, atol=tol)
    assert (np.diff(got_val, axis=1) <= tol).all(), "neighbours are not sorted by descending similarity"
    # (b) no duplicates inside a row
    srt = np.sort(idx, axis=1)
    assert (np.diff(srt, axis=1) > 0).all(), "duplicate neighbour in a row"
    if sims is not None:
        # similarities as REPORTED by the kernel (split-bf16 drops the lo*lo term: <= ~5e-6 low on a self-similarity)
        np.testing.assert_allclose(np.asarray(sims, dtype=np.float64), got_val, rtol=0, atol=sim_tol)
    return int((np.sort(ref_idx, axis=1) != srt).any(axis=1).sum())
