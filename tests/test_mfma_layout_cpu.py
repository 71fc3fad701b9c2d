"""CPU: the numpy model of the MFMA layouts (tools/mfma_layout.py) and the three tricks built on them."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mfma_layout", os.path.join(ROOT, "tools", "mfma_layout.py"))
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)


def test_tile_product_through_the_fragment_layouts():
    g = np.random.default_rng(0)
    A = g.standard_normal((32, 64)).astype(np.float16)
    W = g.standard_normal((32, 64)).astype(np.float16)            # B = W^T
    acc = np.zeros((64, 16), np.float32)
    for ks in range(4):
        acc = L.mfma(L.a_fragment(A[:, 16 * ks: 16 * ks + 16]), L.b_fragment(W[:, 16 * ks: 16 * ks + 16]), acc)
    D = A.astype(np.float32) @ W.astype(np.float32).T
    rows = L.acc_rows()
    assert np.allclose(acc, D[rows, L.LI[:, None]], rtol=1e-5, atol=1e-4)


def test_permuted_key_rows_make_the_scores_a_b_fragment():
    """attention.hip: S^T = K Q^T with the K rows fed in swap_bits23 order; registers 8 s .. 8 s + 7 of a lane then are the B-operand
    fragment (8 consecutive keys of one query column) of k-step s of O^T += V^T P^T - P never crosses lanes."""
    g = np.random.default_rng(1)
    K = g.standard_normal((32, 16)).astype(np.float16)             # 32 keys, one k-step of the head dim
    Q = g.standard_normal((32, 16)).astype(np.float16)             # 32 queries
    Vt = g.standard_normal((32, 32)).astype(np.float16)            # [d, key]
    perm = L.swap_bits23(np.arange(32))
    assert sorted(perm) == list(range(32))
    st = L.mfma(L.a_fragment(K[perm]), L.b_fragment(Q), np.zeros((64, 16), np.float32))      # D row i = key perm[i], column = query
    S = K.astype(np.float32) @ Q.astype(np.float32).T                                        # [key, query]
    for lane in range(64):
        li, lh = lane % 32, lane // 32
        for s in range(2):
            keys = 16 * s + 8 * lh + np.arange(8)
            assert np.allclose(st[lane, 8 * s: 8 * s + 8], S[keys, li], rtol=1e-5, atol=1e-4)
    # so the registers feed the second matmul as they are
    P = st.astype(np.float16)
    o = np.zeros((64, 16), np.float32)
    for s in range(2):
        o = L.mfma(L.a_fragment(Vt[:, 16 * s: 16 * s + 16]), P[:, 8 * s: 8 * s + 8], o)
    O = Vt.astype(np.float32) @ S.astype(np.float16).astype(np.float32)                      # [d, query]
    rows = L.acc_rows()
    assert np.allclose(o, O[rows, L.LI[:, None]], rtol=1e-3, atol=1e-2)


def test_interleaved_weight_rows_give_a_lane_two_adjacent_columns():
    """gemm_kernels.h direct epilogue: with the weight tile's LDS rows in col_map order, a lane's accumulators of the two 32-column blocks
    are columns 2 li and 2 li + 1 - one dword store, 32 lanes cover 128 contiguous bytes of an output row."""
    r = np.arange(64)
    cols = L.col_map(r)
    assert sorted(cols) == list(range(64))
    for li in range(32):
        assert cols[li] == 2 * li and cols[32 + li] == 2 * li + 1
    assert list(L.col_map(np.arange(64, 128))) == [64 + c for c in cols]


def test_single_head_attention_with_128_wide_heads_in_these_layouts():
    """The plan for flow_gmflow's attention (the ViT kernel is built for 64-wide heads): one wave, 32 queries, head
    dim 128 = 8 k-steps of S^T, O^T as four 32-row blocks of d, online softmax over key tiles of 32 - everything a lane needs for its query
    column is in its own registers plus one exchange with lane ^ 32."""
    g = np.random.default_rng(2)
    d, nk = 128, 96
    Q = (g.standard_normal((32, d)) * 0.3).astype(np.float16)
    K = (g.standard_normal((nk, d)) * 0.3).astype(np.float16)
    V = g.standard_normal((nk, d)).astype(np.float16)
    scale = d ** -0.5
    perm = L.swap_bits23(np.arange(32))
    rows = L.acc_rows()
    m = np.full(64, -np.inf, np.float32)                  # per lane: running max of its query column (equal in lanes li and li + 32)
    l = np.zeros(64, np.float32)
    o = [np.zeros((64, 16), np.float32) for _ in range(4)]
    for t in range(nk // 32):
        Kt, Vt = K[32 * t: 32 * t + 32], V[32 * t: 32 * t + 32].T                # Vt [d, key]
        st = np.zeros((64, 16), np.float32)
        for ks in range(d // 16):
            st = L.mfma(L.a_fragment(Kt[perm][:, 16 * ks: 16 * ks + 16]), L.b_fragment(Q[:, 16 * ks: 16 * ks + 16]), st)
        st *= scale
        tmax = st.max(1)
        tmax = np.maximum(tmax, tmax[L.LANES ^ 32])                                # the one cross-lane exchange
        mn = np.maximum(m, tmax)
        corr = np.exp(m - mn)
        p = np.exp(st - mn[:, None])
        psum = p.sum(1)
        l = l * corr + psum + psum[L.LANES ^ 32]
        m = mn
        ph = p.astype(np.float16)
        for b in range(4):                                                         # O^T rows 32 b .. 32 b + 31 of d
            o[b] *= corr[:, None]
            for s in range(2):
                o[b] = L.mfma(L.a_fragment(Vt[32 * b: 32 * b + 32, 16 * s: 16 * s + 16]), ph[:, 8 * s: 8 * s + 8], o[b])
    S = (Q.astype(np.float32) @ K.astype(np.float32).T) * scale
    P = np.exp(S - S.max(1, keepdims=True))
    ref = (P / P.sum(1, keepdims=True)) @ V.astype(np.float32)                     # [query, d]
    for b in range(4):
        got = o[b] / l[:, None]                                                    # lane (li, lh), register r: d = 32 b + rows, query li
        assert np.allclose(got, ref[L.LI[:, None], 32 * b + rows], rtol=5e-3, atol=5e-3), b
