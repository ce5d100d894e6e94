"""Per-kernel parity: every HIP kernel, called through the C ABI, against the CPU oracle on the same
seeded inputs.  fp32 tolerances are written next to each check (the kernels and the oracle sum in
different orders, so agreement is to rounding, not bitwise, except where noted)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import numpy_ref as R
from oracle import torch_ref as T


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd import _lib
    return _lib


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def st():
    return torch.cuda.current_stream().cuda_stream


def close(got, want, rtol, atol):
    np.testing.assert_allclose(got.detach().cpu().double().numpy(), np.asarray(want, np.float64), rtol=rtol, atol=atol)


WS = 8 * 1024 * 1024


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (6144, 728, 728), (100, 728, 256), (37, 64, 288), (32, 576, 4096),
                                   (1000, 128, 64), (4, 12, 8)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_gemm_forward_form(L, M, N, K, tile):
    rs = np.random.RandomState(M + N + K)
    A, B = rs.randn(M, K).astype(np.float32), rs.randn(K, N).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    a, b, bi = dev(A), dev(B), dev(bias)
    c = torch.full((M, N), float("nan"), device="cuda")
    ws = torch.empty(WS, device="cuda")
    L.spnet_gemm_f32(a.data_ptr(), 0, K, b.data_ptr(), 1, N, c.data_ptr(), N, M, N, K, 0, ws.data_ptr(), WS,
                     bi.data_ptr(), tile, st())
    want = A.astype(np.float64) @ B.astype(np.float64) + bias
    # f32 fma chain over K terms of magnitude ~1: error ~ 1e-7*sqrt(K)*|terms|
    close(c, want, rtol=2e-5, atol=2e-5 * np.sqrt(K))


@pytest.mark.parametrize("split", [1, 2, 7])
def test_gemm_split_k_is_deterministic_and_correct(L, split):
    rs = np.random.RandomState(3)
    M, N, K = 96, 64, 1000
    A, B = rs.randn(M, K).astype(np.float32), rs.randn(K, N).astype(np.float32)
    a, b = dev(A), dev(B)
    ws = torch.empty(WS, device="cuda")
    outs = []
    for _ in range(2):
        c = torch.empty(M, N, device="cuda")
        L.spnet_gemm_f32(a.data_ptr(), 0, K, b.data_ptr(), 1, N, c.data_ptr(), N, M, N, K, split, ws.data_ptr(), WS,
                         None, 3, st())
        outs.append(c.cpu())
    assert torch.equal(outs[0], outs[1])
    close(outs[0], A.astype(np.float64) @ B.astype(np.float64), rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(6144, 728, 728), (200, 64, 128), (32, 4096, 576)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_gemm_dgrad_form(L, M, N, K, tile):
    # dX[M,N=cin] = dY[M,K=cout] @ W[N,K]^T, W read in place (K-major B)
    rs = np.random.RandomState(1)
    dY, W = rs.randn(M, K).astype(np.float32), rs.randn(N, K).astype(np.float32)
    a, b = dev(dY), dev(W)
    c = torch.empty(M, N, device="cuda")
    ws = torch.empty(WS, device="cuda")
    c.fill_(float("nan"))
    L.spnet_gemm_f32(a.data_ptr(), 0, K, b.data_ptr(), 0, K, c.data_ptr(), N, M, N, K, 0, ws.data_ptr(), WS, None, tile, st())
    close(c, dY.astype(np.float64) @ W.astype(np.float64).T, rtol=2e-5, atol=2e-5 * np.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(728, 728, 6144), (64, 128, 23250), (4096, 576, 32), (288, 64, 5000)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5, 6, 7, 8, 9, 10])
def test_gemm_wgrad_form(L, M, N, K, tile):
    # dW[M=cin,N=cout] = X[K,M]^T @ dY[K,N]; K (pixels) need not be a multiple of 4
    rs = np.random.RandomState(2)
    X, dY = rs.randn(K, M).astype(np.float32), rs.randn(K, N).astype(np.float32)
    a, b = dev(X), dev(dY)
    c = torch.empty(M, N, device="cuda")
    ws = torch.empty(WS, device="cuda")
    c.fill_(float("nan"))
    L.spnet_gemm_f32(a.data_ptr(), 1, M, b.data_ptr(), 1, N, c.data_ptr(), N, M, N, K, 0, ws.data_ptr(), WS, None, tile, st())
    close(c, X.astype(np.float64).T @ dY.astype(np.float64), rtol=2e-5, atol=3e-5 * np.sqrt(K))


@pytest.mark.parametrize("tile", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("split", [1, 2])
@pytest.mark.parametrize("form", ["fwd", "dgrad", "wgrad"])
def test_gemm_long_k_takes_the_pipelined_main_loop(L, form, tile, split):
    """From 32 K tiles per workgroup the launcher picks the explicitly software-pipelined main loop (gemm.hip, PIPE):
    all three operand forms with ragged M / N / K, unsplit and with a K split whose last slice is short."""
    rs = np.random.RandomState(tile + 10 * split)
    ws = torch.empty(WS, device="cuda")
    if form == "fwd":          # A[M,K] K-major, B[K,N]
        M, N, K = 100, 72, 4100
        A, B = rs.randn(M, K).astype(np.float32), rs.randn(K, N).astype(np.float32)
        a, b, args = dev(A), dev(B), (0, K, 1, N)
        want = A.astype(np.float64) @ B.astype(np.float64)
    elif form == "dgrad":      # B[N,K] K-major
        M, N, K = 200, 68, 2180
        A, B = rs.randn(M, K).astype(np.float32), rs.randn(N, K).astype(np.float32)
        a, b, args = dev(A), dev(B), (0, K, 0, K)
        want = A.astype(np.float64) @ B.astype(np.float64).T
    else:                      # A[K,M] out-major, K not a multiple of 4
        M, N, K = 72, 136, 5001
        A, B = rs.randn(K, M).astype(np.float32), rs.randn(K, N).astype(np.float32)
        a, b, args = dev(A), dev(B), (1, M, 1, N)
        want = A.astype(np.float64).T @ B.astype(np.float64)
    c = torch.full((M, N), float("nan"), device="cuda")
    L.spnet_gemm_f32(a.data_ptr(), args[0], args[1], b.data_ptr(), args[2], args[3], c.data_ptr(), N, M, N, K, split,
                     ws.data_ptr(), WS, None, tile, st())
    close(c, want, rtol=2e-5, atol=3e-5 * np.sqrt(K))


@pytest.mark.parametrize("M,N,K,form", [(9744, 320, 32, "dgrad"), (2240, 1088, 192, "dgrad"), (100, 72, 40, "fwd"),
                                        (384, 2080, 192, "dgrad"), (37, 64, 288, "fwd")])
@pytest.mark.parametrize("tile", [0, 3, 6])
def test_gemm_accumulate_into_c(L, M, N, K, form, tile):
    """spnet_gemm_f32_accumulate: C += A B in the epilogue == the GEMM into a scratch tensor followed by an elementwise
    add, bit for bit (one rounding of the product sum, one of the addition, either way)."""
    rs = np.random.RandomState(M + N + K)
    A = dev(rs.randn(M, K))
    if form == "fwd":
        B, bmaj, ldb = dev(rs.randn(K, N) * 0.1), 1, N
    else:
        B, bmaj, ldb = dev(rs.randn(N, K) * 0.1), 0, K
    C0 = dev(rs.randn(M, N))
    scratch = torch.empty(M, N, device="cuda")
    L.spnet_gemm_f32(A.data_ptr(), 0, K, B.data_ptr(), bmaj, ldb, scratch.data_ptr(), N, M, N, K, 1, None, 0, None, tile, st())
    want = scratch + C0
    C = C0.clone()
    L.spnet_gemm_f32_accumulate(A.data_ptr(), 0, K, B.data_ptr(), bmaj, ldb, C.data_ptr(), N, M, N, K, tile, st())
    assert torch.equal(C, want)


def x3_planes(L, R, K, fill=0):
    """a zeroed (or poisoned) plane set for an [R][K] matrix"""
    n = 3 * int(L.spnet_bf16x3_plane_elems(R, K))
    return torch.full((n,), fill, dtype=torch.int16, device="cuda")


def x3_untile(planes, R, K):
    """planes in the pieces layout of csrc/x3t.h -> float64 [3][R16][Kp] (R16, Kp: R, K rounded up to 16 / 32): element
    (r, k) of a plane sits at ((r/16)*nk + k/32)*512 + (r%16)*32 + (((k%32)/8 ^ -((r%16)/4)) & 3)*8 + k%8."""
    nk, rg = (K + 31) // 32, (R + 15) // 16
    p = planes.view(torch.bfloat16).reshape(3, rg, nk, 16, 4, 8).float().cpu().double()
    r16 = torch.arange(16)
    cpos = torch.arange(4)[None, :] ^ ((-(r16 // 4)) & 3)[:, None]            # [r16][chunk] -> stored position
    q = p[:, :, :, r16[:, None], cpos, :]                                     # [3][rg][nk][16][chunk][8]
    return q.permute(0, 1, 3, 2, 4, 5).reshape(3, rg * 16, nk * 32)


@pytest.mark.parametrize("M,N,K", [(6144, 728, 728), (200, 96, 32), (97, 100, 260), (1536, 1024, 1536), (5000, 800, 100),
                                   (300, 128, 160), (300, 128, 192)])
def test_gemm_bf16x3_accuracy(L, M, N, K):
    """The bf16x3 kernels (fp32 operands as three bf16 pieces, six bf16 MFMAs, fp32 accumulation; the pointwise GEMMs of the
    product path): the error against float64 is of the size of the exact fp32 kernel's own -- within 3x of it, and below 2e-6
    of the row / column norms product -- on random operands with a wide dynamic range; the planes reproduce their matrix
    exactly and their pad rows / columns are zero; the planes x planes kernel (A split by its producer) and the kernel that
    splits an fp32 A while staging it give the same bits."""
    rs = np.random.RandomState(M + N)
    A = (rs.randn(M, K) * np.exp(rs.randn(M, K))).astype(np.float32)
    W = (rs.randn(K, N) * 0.1 * np.exp(rs.randn(K, N))).astype(np.float32)
    a, w = dev(A), dev(W)
    planes = x3_planes(L, N, K)
    L.spnet_split_bf16x3(w.data_ptr(), planes.data_ptr(), K, N, st())
    pl = x3_untile(planes, N, K)
    assert torch.equal((pl[0] + pl[1] + pl[2])[:N, :K].T.contiguous(), torch.from_numpy(W).double())     # h + m + l == w exactly
    assert float(pl[:, N:].abs().max() if pl.shape[1] > N else 0) == 0.0 and float(pl[:, :, K:].abs().max() if pl.shape[2] > K else 0) == 0.0
    apl = x3_planes(L, M, K)
    L.spnet_split_rows_bf16x3(a.data_ptr(), K, apl.data_ptr(), M, K, st())
    al = x3_untile(apl, M, K)
    assert torch.equal((al[0] + al[1] + al[2])[:M, :K], torch.from_numpy(A).double())
    c3 = torch.full((M, N), float("nan"), device="cuda")
    L.spnet_gemm_bf16x3_fwd(a.data_ptr(), K, planes.data_ptr(), c3.data_ptr(), N, M, N, K, st())
    cp = torch.full((M + 1, N), 7.0, device="cuda")                     # (a guard row behind C)
    L.spnet_gemm_bf16x3_pp(apl.data_ptr(), planes.data_ptr(), cp.data_ptr(), N, M, N, K, None, None, st())
    assert torch.equal(cp[:M], c3) and bool((cp[M] == 7.0).all())
    c1 = torch.empty(M, N, device="cuda")
    L.spnet_gemm_f32(a.data_ptr(), 0, K, w.data_ptr(), 1, N, c1.data_ptr(), N, M, N, K, 1, None, 0, None, 0, st())
    ref = A.astype(np.float64) @ W.astype(np.float64)
    scale = np.sqrt((A.astype(np.float64) ** 2).sum(1))[:, None] * np.sqrt((W.astype(np.float64) ** 2).sum(0))[None, :]
    e3 = np.abs(c3.cpu().double().numpy() - ref) / scale
    e1 = np.abs(c1.cpu().double().numpy() - ref) / scale
    assert np.isfinite(e3).all()
    assert e3.max() < 2e-6 and e3.max() < 3.0 * max(e1.max(), 1e-8), (e3.max(), e1.max())
    assert np.sqrt((e3 ** 2).mean()) < 3.0 * max(np.sqrt((e1 ** 2).mean()), 1e-9)


def test_gemm_bf16x3_nan_and_inf_propagate(L):
    """A NaN or an infinity in an operand must come out as non-finite results, as on the exact chain (a split by integer
    rounding turns some NaNs into zeros: MI355X_MICROARCH.md, correctness boundaries): every split goes through
    v_cvt_pk_bf16_f32."""
    M, N, K = 96, 96, 64
    rs = np.random.RandomState(0)
    A, W = rs.randn(M, K).astype(np.float32), rs.randn(K, N).astype(np.float32)
    A[3, 5] = np.float32(np.nan)
    A[7, 9] = np.frombuffer(np.uint32(0x7FFFFFFF).tobytes(), np.float32)[0]          # NaN with every payload bit set
    W[11, 13] = np.float32(np.inf)
    W[20, 21] = np.frombuffer(np.uint32(0xFFFFFFFF).tobytes(), np.float32)[0]
    a, w = dev(A), dev(W)
    planes, apl = x3_planes(L, N, K), x3_planes(L, M, K)
    L.spnet_split_bf16x3(w.data_ptr(), planes.data_ptr(), K, N, st())
    L.spnet_split_rows_bf16x3(a.data_ptr(), K, apl.data_ptr(), M, K, st())
    for kind in ("fwd", "pp"):
        c = torch.zeros(M, N, device="cuda")
        if kind == "fwd":
            L.spnet_gemm_bf16x3_fwd(a.data_ptr(), K, planes.data_ptr(), c.data_ptr(), N, M, N, K, st())
        else:
            L.spnet_gemm_bf16x3_pp(apl.data_ptr(), planes.data_ptr(), c.data_ptr(), N, M, N, K, None, None, st())
        c = c.cpu().numpy()
        assert np.isnan(c[3]).all() and np.isnan(c[7]).all(), kind          # NaN rows of A
        assert not np.isfinite(c[:, 13]).any() and np.isnan(c[:, 21]).all(), kind     # inf / NaN columns of W
        ok = np.ones((M, N), bool)
        ok[[3, 7]] = False
        ok[:, [13, 21]] = False
        assert np.isfinite(c[ok]).all(), kind


@pytest.mark.parametrize("M,N,K", [(6144, 728, 728), (5483, 732, 260), (1000, 256, 128), (97, 260, 100)])
def test_gemm_bf16x3_colstats_batched_split_and_dgrad_form(L, M, N, K):
    """The pieces the engine uses: spnet_split_bf16x3_batched makes the planes of both operand forms in one launch
    (identical to the single split for the forward form); the forward GEMM with BatchNorm column sums leaves exactly the
    sums of the C it wrote, tile row by tile row (both operand forms of A); the data-gradient form dX = dY W^T on the planes
    of W as stored."""
    import ctypes
    rs = np.random.RandomState(K)
    A = (rs.randn(M, K)).astype(np.float32)
    W = (rs.randn(K, N) * 0.1).astype(np.float32)
    G = (rs.randn(M, N)).astype(np.float32)
    a, w, g = dev(A), dev(W), dev(G)
    pe = lambda r, k: int(L.spnet_bf16x3_plane_elems(r, k))
    pf1 = x3_planes(L, N, K)
    L.spnet_split_bf16x3(w.data_ptr(), pf1.data_ptr(), K, N, st())
    pf, pd = x3_planes(L, N, K, -1), x3_planes(L, K, N, -1)       # (poisoned: the split writes every element, pads included)
    jobs = torch.tensor([w.data_ptr(), pf.data_ptr(), K, N, 1, N, w.data_ptr(), pd.data_ptr(), N, K, N, 1], dtype=torch.int64,
                        device="cuda")
    L.spnet_split_bf16x3_batched(jobs.data_ptr(), 2, max(pe(N, K), pe(K, N)), st())
    assert torch.equal(pf, pf1)
    pl = x3_untile(pd, K, N)
    assert torch.equal((pl[0] + pl[1] + pl[2])[:K, :N], torch.from_numpy(W).double())          # element (n=k_in, k=n_out) = W[n][k]
    assert float(pl[:, K:].abs().max() if pl.shape[1] > K else 0) == 0.0 and float(pl[:, :, N:].abs().max() if pl.shape[2] > N else 0) == 0.0
    # forward + column sums
    c = torch.full((M, N), float("nan"), device="cuda")
    rows = ctypes.c_int(0)
    cs = torch.full(((M + 95) // 96 * 2 * N,), float("nan"), device="cuda")
    L.spnet_gemm_bf16x3_fwd_colstats(a.data_ptr(), K, pf.data_ptr(), c.data_ptr(), N, M, N, K, cs.data_ptr(),
                                     ctypes.addressof(rows), st())
    assert rows.value == (M + 95) // 96
    c0 = torch.empty(M, N, device="cuda")
    L.spnet_gemm_bf16x3_fwd(a.data_ptr(), K, pf.data_ptr(), c0.data_ptr(), N, M, N, K, st())
    assert torch.equal(c, c0)
    close(c, A.astype(np.float64) @ W.astype(np.float64), rtol=2e-5, atol=2e-5 * np.sqrt(K))
    part = cs.reshape(rows.value, 2, N).cpu().double().numpy()
    cn = c.cpu().double().numpy()
    for t in range(rows.value):
        blk = cn[t * 96:(t + 1) * 96]
        np.testing.assert_allclose(part[t, 0], blk.sum(0), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(part[t, 1], (blk ** 2).sum(0), rtol=1e-5, atol=1e-4)
    # the same from A's planes: C and the partial sums bit for bit
    apl = x3_planes(L, M, K)
    L.spnet_split_rows_bf16x3(a.data_ptr(), K, apl.data_ptr(), M, K, st())
    c2 = torch.full((M, N), float("nan"), device="cuda")
    cs2 = torch.full_like(cs, float("nan"))
    rows2 = ctypes.c_int(0)
    L.spnet_gemm_bf16x3_pp(apl.data_ptr(), pf.data_ptr(), c2.data_ptr(), N, M, N, K, cs2.data_ptr(), ctypes.addressof(rows2), st())
    assert rows2.value == rows.value and torch.equal(c2, c) and torch.equal(cs2, cs)
    # data gradient: dX[M][K] = G[M][N] W^T
    dx = torch.full((M, K), float("nan"), device="cuda")
    L.spnet_gemm_bf16x3_fwd(g.data_ptr(), N, pd.data_ptr(), dx.data_ptr(), K, M, K, N, st())
    close(dx, G.astype(np.float64) @ W.astype(np.float64).T, rtol=2e-5, atol=2e-5 * np.sqrt(N))
    gpl = x3_planes(L, M, N)
    L.spnet_split_rows_bf16x3(g.data_ptr(), N, gpl.data_ptr(), M, N, st())
    dx2 = torch.full((M, K), float("nan"), device="cuda")
    L.spnet_gemm_bf16x3_pp(gpl.data_ptr(), pd.data_ptr(), dx2.data_ptr(), K, M, K, N, None, None, st())
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("M,cin,cout,nb", [(6144, 728, 728, 2), (1000, 100, 260, 3), (96, 96, 96, 1), (1552, 256, 728, 1),
                                           (5000, 260, 36, 2)])
def test_gemm_bf16x3_wgrad_from_planes(L, M, cin, cout, nb):
    """spnet_gemm_bf16x3_wgrad_batched: dW = z^T dy from the planes the forward / data-gradient launches read (fragments by
    transposing LDS reads), several problems per launch; error against float64 of the size of the exact fp32 kernel's own;
    with a K split + spnet_reduce_slabs the same to rounding, and bit-identical from run to run; pixel counts that are not a
    multiple of 32 (a row group missing from the last contraction step) and channel counts that are not a multiple of 96."""
    rs = np.random.RandomState(M + cin)
    Z = [(rs.randn(M, cin) * np.exp(0.5 * rs.randn(M, cin))).astype(np.float32) for _ in range(nb)]
    G = [(rs.randn(M, cout) * 0.1).astype(np.float32) for _ in range(nb)]
    zp, gp = [x3_planes(L, M, cin) for _ in range(nb)], [x3_planes(L, M, cout) for _ in range(nb)]
    zd, gd = [dev(z) for z in Z], [dev(g) for g in G]
    for b in range(nb):
        L.spnet_split_rows_bf16x3(zd[b].data_ptr(), cin, zp[b].data_ptr(), M, cin, st())
        L.spnet_split_rows_bf16x3(gd[b].data_ptr(), cout, gp[b].data_ptr(), M, cout, st())
    dw = [torch.full((cin + 1, cout), 7.0, device="cuda") for _ in range(nb)]
    jobs = torch.tensor([v for b in range(nb) for v in (zp[b].data_ptr(), gp[b].data_ptr(), dw[b].data_ptr())], dtype=torch.int64,
                        device="cuda")
    L.spnet_gemm_bf16x3_wgrad_batched(jobs.data_ptr(), nb, cin, cout, M, 1, st())
    for b in range(nb):
        ref = Z[b].astype(np.float64).T @ G[b].astype(np.float64)
        scale = np.sqrt((Z[b].astype(np.float64) ** 2).sum(0))[:, None] * np.sqrt((G[b].astype(np.float64) ** 2).sum(0))[None, :]
        c1 = torch.empty(cin, cout, device="cuda")
        L.spnet_gemm_f32(zd[b].data_ptr(), 1, cin, gd[b].data_ptr(), 1, cout, c1.data_ptr(), cout, cin, cout, M, 1, None, 0, None, 0, st())
        e3 = np.abs(dw[b][:cin].cpu().double().numpy() - ref) / scale
        e1 = np.abs(c1.cpu().double().numpy() - ref) / scale
        assert bool((dw[b][cin] == 7.0).all())
        assert np.isfinite(e3).all() and e3.max() < 2e-6 and e3.max() < 3.0 * max(e1.max(), 1e-8), (b, e3.max(), e1.max())
    # K split: slabs + ordered sum
    ks = max(2, min(7, ((M + 31) // 32) // 3))
    per = ((M + 31) // 32 + ks - 1) // ks
    ks = ((M + 31) // 32 + per - 1) // per
    if ks > 1:
        slabs = [torch.full((ks, cin, cout), float("nan"), device="cuda") for _ in range(nb)]
        jobs2 = torch.tensor([v for b in range(nb) for v in (zp[b].data_ptr(), gp[b].data_ptr(), slabs[b].data_ptr())],
                             dtype=torch.int64, device="cuda")
        outs = []
        for rep in range(2):
            for sl in slabs:
                sl.fill_(float("nan"))
            L.spnet_gemm_bf16x3_wgrad_batched(jobs2.data_ptr(), nb, cin, cout, M, ks, st())
            o = [torch.empty(cin, cout, device="cuda") for _ in range(nb)]
            for b in range(nb):
                L.spnet_reduce_slabs(slabs[b].data_ptr(), ks, cin, cout, o[b].data_ptr(), cout, st())
            outs.append(o)
        for b in range(nb):
            assert torch.equal(outs[0][b], outs[1][b])
            ref = Z[b].astype(np.float64).T @ G[b].astype(np.float64)
            close(outs[0][b], ref, rtol=2e-5, atol=2e-5 * np.sqrt(M) * 0.3)
    assert int(L.spnet_gemm_bf16x3_wgrad_ksplit(256, 256, 94752, 1)) > 8 and int(L.spnet_gemm_bf16x3_wgrad_ksplit(728, 728, 6144, 24)) == 1


def _planes_equal_split_of(L, planes, y, M, C):
    """`planes` (written by a producer kernel) == the split of the fp32 tensor y [M][C] the fp32 form of that kernel
    writes: bit for bit, pads included (they must still be zero), and h + m + l == y exactly."""
    ref = x3_planes(L, M, C)
    L.spnet_split_rows_bf16x3(y.data_ptr(), C, ref.data_ptr(), M, C, st())
    assert torch.equal(planes, ref)
    pl = x3_untile(planes, M, C)
    assert torch.equal((pl[0] + pl[1] + pl[2])[:M, :C], y.reshape(M, C).cpu().double())


@pytest.mark.parametrize("B,H,W,C", [(2, 12, 16, 728), (3, 7, 5, 64), (2, 6, 8, 1536), (1, 1, 1, 8), (2, 24, 32, 256),
                                     (1, 47, 63, 128), (2, 13, 17, 40)])
@pytest.mark.parametrize("form", ["tiled", "stream", "stream5", "bnfin"])
def test_dwconv_forward_writes_bf16x3_planes(L, B, H, W, C, form):
    """spnet_dwconv3x3_{tiled,stream}_fwd_x3 / _tiled_fwd_bnfin_x3: the depthwise output as the bf16x3 planes the pointwise
    GEMMs read == the split of the fp32 form's output, bit for bit (same arithmetic, another store)."""
    rs = np.random.RandomState(B * H + C)
    x = dev(rs.randn(B, H, W, C) * np.exp(rs.randn(B, H, W, C)))
    w = dev(rs.randn(3, 3, C))
    sc, sh = dev(rs.rand(C) + 0.5), dev(rs.randn(C) * 0.1)
    M = B * H * W
    y = torch.full((B, H, W, C), float("nan"), device="cuda")
    planes = x3_planes(L, M, C)
    if form == "bnfin":
        rows = 5
        part = dev(rs.rand(rows, 2, C) * M)
        part[:, 1] += part[:, 0] ** 2 / M          # sum of squares >= (sum)^2 / M
        gam, bet = dev(rs.rand(C) + 0.5), dev(rs.randn(C) * 0.1)
        outs = []
        for k in range(2):
            mm, mv, sm, si, ss = [torch.zeros(C, device="cuda") for _ in range(4)] + [torch.zeros(2 * C, device="cuda")]
            args = (B, H, W, C, 1, part.data_ptr(), rows, M, gam.data_ptr(), bet.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                    sm.data_ptr(), si.data_ptr(), ss.data_ptr(), 1e-3, 0.99, st())
            if k == 0:
                L.spnet_dwconv3x3_tiled_fwd_bnfin(x.data_ptr(), w.data_ptr(), y.data_ptr(), *args)
            else:
                L.spnet_dwconv3x3_tiled_fwd_bnfin_x3(x.data_ptr(), w.data_ptr(), planes.data_ptr(), *args)
            outs.append((mm, mv, sm, si, ss))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    elif form == "tiled":
        L.spnet_dwconv3x3_tiled_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, C, 1, sc.data_ptr(), sh.data_ptr(), st())
        L.spnet_dwconv3x3_tiled_fwd_x3(x.data_ptr(), w.data_ptr(), planes.data_ptr(), B, H, W, C, 1, sc.data_ptr(), sh.data_ptr(), st())
    else:
        rps = 5 if form == "stream5" else 0
        L.spnet_dwconv3x3_stream_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, C, 1, sc.data_ptr(), sh.data_ptr(), rps, st())
        L.spnet_dwconv3x3_stream_fwd_x3(x.data_ptr(), w.data_ptr(), planes.data_ptr(), B, H, W, C, 1, sc.data_ptr(), sh.data_ptr(), rps, st())
    assert bool(torch.isfinite(y).all())
    _planes_equal_split_of(L, planes, y, M, C)


@pytest.mark.parametrize("M,C,P", [(6144, 728, 32), (6144, 728, 160), (1536, 1536, 16), (500, 36, 8), (94752, 256, 128)])
@pytest.mark.parametrize("act", [0, 1])
def test_batchnorm_backward_writes_bf16x3_planes(L, M, C, P, act):
    """spnet_bn_bwd_from_partials_x3 (both of its paths: one launch up to 128 partial rows, finalize + apply beyond) and
    spnet_bn_bwd_x3: dx as bf16x3 planes == the split of the fp32 form's dx, bit for bit; dgamma / dbeta identical."""
    rs = np.random.RandomState(M + C)
    x, g = dev(rs.randn(M, C)), dev(rs.randn(M, C) * np.exp(rs.randn(M, C)))
    gamma, beta = dev(rs.rand(C) + 0.5), dev(rs.randn(C) * 0.1)
    mu, isd = dev(rs.randn(C) * 0.1), dev(rs.rand(C) + 0.5)
    part = dev(rs.randn(P, 2, C))
    co = torch.empty(3 * C, device="cuda")
    if act == 0:
        dx = torch.full((M, C), float("nan"), device="cuda")
        d0, d1 = [torch.empty(C, device="cuda") for _ in range(2)], [torch.empty(C, device="cuda") for _ in range(2)]
        planes = x3_planes(L, M, C)
        a = (x.data_ptr(), g.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), mu.data_ptr(), isd.data_ptr(), P, part.data_ptr())
        L.spnet_bn_bwd_from_partials(*a, dx.data_ptr(), d0[0].data_ptr(), d0[1].data_ptr(), co.data_ptr(), st())
        L.spnet_bn_bwd_from_partials_x3(*a, planes.data_ptr(), d1[0].data_ptr(), d1[1].data_ptr(), co.data_ptr(), st())
        assert torch.equal(d0[0], d1[0]) and torch.equal(d0[1], d1[1])
        _planes_equal_split_of(L, planes, dx, M, C)
    ws = torch.empty(L.spnet_bn_ws(M, C), device="cuda")
    dx = torch.full((M, C), float("nan"), device="cuda")
    d0, d1 = [torch.empty(C, device="cuda") for _ in range(2)], [torch.empty(C, device="cuda") for _ in range(2)]
    planes = x3_planes(L, M, C)
    a = (x.data_ptr(), g.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), mu.data_ptr(), isd.data_ptr(), act)
    L.spnet_bn_bwd(*a, dx.data_ptr(), d0[0].data_ptr(), d0[1].data_ptr(), co.data_ptr(), ws.data_ptr(), st())
    L.spnet_bn_bwd_x3(*a, planes.data_ptr(), d1[0].data_ptr(), d1[1].data_ptr(), co.data_ptr(), ws.data_ptr(), st())
    assert torch.equal(d0[0], d1[0]) and torch.equal(d0[1], d1[1])
    _planes_equal_split_of(L, planes, dx, M, C)


@pytest.mark.parametrize("B,H,W,cin,cout", [(32, 12, 16, 728, 728), (3, 12, 16, 256, 260), (5, 6, 8, 1024, 1536), (2, 6, 8, 100, 96),
                                            (1, 12, 16, 96, 32), (7, 6, 8, 36, 64)])
@pytest.mark.parametrize("with_add,with_bnx", [(0, 0), (1, 0), (1, 1), (0, 1)])
def test_gemm_bf16x3_dgrad_with_fused_depthwise_backward(L, B, H, W, cin, cout, with_add, with_bnx):
    """spnet_gemm_bf16x3_pp_dwbwd: the pointwise data-gradient GEMM with the depthwise backward in its epilogue == the two
    launches it replaces (spnet_gemm_bf16x3_pp into dz, then spnet_dwconv3x3_tiled_bwd over dz): dx bit for bit, the
    depthwise weight-gradient sums and the producer BatchNorm's backward sums (partial rows group the pixels differently)
    to rounding; batch sizes whose last 192-pixel tile is ragged, channel counts that are not multiples of 96."""
    assert L.spnet_gemm_bf16x3_dwbwd_ok(H, W, cin) == 1 and L.spnet_gemm_bf16x3_dwbwd_ok(24, 32, cin) == 0
    assert L.spnet_gemm_bf16x3_dwbwd_ok(3, 4, cin) == 0
    rs = np.random.RandomState(B * H + cin + cout)
    M = B * H * W
    dy = dev(rs.randn(M, cout) * 0.5)
    wpw = dev(rs.randn(cin, cout) * 0.1)
    x = dev(rs.randn(B, H, W, cin))
    wd = dev(rs.randn(3, 3, cin))
    add = dev(rs.randn(B, H, W, cin)) if with_add else None
    bnx = dev(rs.randn(B, H, W, cin)) if with_bnx else None
    sc, sh = dev(rs.rand(cin) + 0.5), dev(rs.randn(cin) * 0.3)
    mu, isd = dev(rs.randn(cin) * 0.1), dev(rs.rand(cin) + 0.5)
    dyp, wp = x3_planes(L, M, cout), x3_planes(L, cin, cout)
    L.spnet_split_rows_bf16x3(dy.data_ptr(), cout, dyp.data_ptr(), M, cout, st())
    jobs = torch.tensor([wpw.data_ptr(), wp.data_ptr(), cout, cin, cout, 1], dtype=torch.int64, device="cuda")    # W as stored
    L.spnet_split_bf16x3_batched(jobs.data_ptr(), 1, int(L.spnet_bf16x3_plane_elems(cin, cout)), st())
    # reference path: dz = dy W^T, then the tile kernel
    dz = torch.empty(M, cin, device="cuda")
    L.spnet_gemm_bf16x3_pp(dyp.data_ptr(), wp.data_ptr(), dz.data_ptr(), cin, M, cin, cout, None, None, st())
    rows0 = int(L.spnet_dwconv3x3_tiled_rows(B, H, W, cin))
    ws0 = torch.zeros(int(L.spnet_dwconv3x3_tiled_bwd_ws(B, H, W, cin)), device="cuda")
    bnp0 = torch.zeros(rows0 * 2 * cin, device="cuda")
    dx0 = torch.full((B, H, W, cin), float("nan"), device="cuda")
    p = lambda t: None if t is None else t.data_ptr()
    L.spnet_dwconv3x3_tiled_bwd(dz.data_ptr(), x.data_ptr(), wd.data_ptr(), dx0.data_ptr(), None, B, H, W, cin, 1, p(add),
                                ws0.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), bnp0.data_ptr(), p(bnx), st())
    # fused
    rows1 = int(L.spnet_gemm_bf16x3_dwbwd_rows(M))
    assert rows1 == (M + 191) // 192
    part1 = torch.full((rows1 * 9 * cin,), float("nan"), device="cuda")
    bnp1 = torch.full((rows1 * 2 * cin,), float("nan"), device="cuda")
    dx1 = torch.full((B * H * W + 1, cin), 7.0, device="cuda")
    L.spnet_gemm_bf16x3_pp_dwbwd(dyp.data_ptr(), wp.data_ptr(), B, H, W, cin, cout, x.data_ptr(), wd.data_ptr(), dx1.data_ptr(), 1,
                                 p(add), part1.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(),
                                 bnp1.data_ptr(), p(bnx), st())
    torch.cuda.synchronize()
    assert torch.equal(dx1[:M].reshape(B, H, W, cin), dx0) and bool((dx1[M] == 7.0).all())
    dk0 = ws0[:rows0 * 9 * cin].reshape(rows0, 9, cin).double().sum(0)
    dk1 = part1.reshape(rows1, 9, cin).double().sum(0)
    scale = float(dk0.abs().max())
    assert float((dk1 - dk0).abs().max()) <= 2e-5 * scale * np.sqrt(M / 192.0)
    b0 = bnp0.reshape(rows0, 2, cin).double().sum(0)
    b1 = bnp1.reshape(rows1, 2, cin).double().sum(0)
    assert float((b1 - b0).abs().max()) <= 2e-5 * float(b0.abs().max()) * np.sqrt(M / 192.0)
    # relu_in = 0 and no BatchNorm sums (the exit flow's first unit): dx again bit for bit
    dx0.fill_(float("nan")); dx1.fill_(7.0)
    L.spnet_dwconv3x3_tiled_bwd(dz.data_ptr(), x.data_ptr(), wd.data_ptr(), dx0.data_ptr(), None, B, H, W, cin, 0, p(add),
                                ws0.data_ptr(), None, None, None, None, None, None, st())
    L.spnet_gemm_bf16x3_pp_dwbwd(dyp.data_ptr(), wp.data_ptr(), B, H, W, cin, cout, x.data_ptr(), wd.data_ptr(), dx1.data_ptr(), 0,
                                 p(add), part1.data_ptr(), None, None, None, None, None, None, st())
    assert torch.equal(dx1[:M].reshape(B, H, W, cin), dx0)


@pytest.mark.parametrize("B,H,W,cin,cout", [(128, 12, 16, 728, 728), (3, 12, 16, 256, 260), (5, 6, 8, 1024, 1536), (7, 6, 8, 36, 64)])
@pytest.mark.parametrize("affine,relu", [(1, 1), (0, 1), (1, 0)])
def test_gemm_bf16x3_forward_with_the_next_depthwise_fused(L, B, H, W, cin, cout, affine, relu):
    """spnet_gemm_bf16x3_pp_dwfwd (inference): pointwise GEMM + folded BatchNorm + ReLU + the next unit's depthwise in the
    epilogue == spnet_gemm_bf16x3_pp into yp followed by spnet_dwconv3x3_stream_fwd_x3(yp): the planes bit for bit."""
    rs = np.random.RandomState(B * H + cin + cout)
    M = B * H * W
    z = dev(rs.randn(M, cin))
    wpw = dev(rs.randn(cin, cout) * 0.1)
    wd = dev(rs.randn(3, 3, cout))
    ss = dev(np.concatenate([rs.rand(cout) + 0.5, rs.randn(cout) * 0.3]))
    zp, wp = x3_planes(L, M, cin), x3_planes(L, cout, cin)
    L.spnet_split_rows_bf16x3(z.data_ptr(), cin, zp.data_ptr(), M, cin, st())
    L.spnet_split_bf16x3(wpw.data_ptr(), wp.data_ptr(), cin, cout, st())
    yp = torch.empty(M, cout, device="cuda")
    L.spnet_gemm_bf16x3_pp(zp.data_ptr(), wp.data_ptr(), yp.data_ptr(), cout, M, cout, cin, None, None, st())
    ref = x3_planes(L, M, cout)
    L.spnet_dwconv3x3_stream_fwd_x3(yp.data_ptr(), wd.data_ptr(), ref.data_ptr(), B, H, W, cout, relu, ss.data_ptr() if affine else None,
                                    ss[cout:].data_ptr() if affine else None, 0, st())
    out = x3_planes(L, M, cout)
    L.spnet_gemm_bf16x3_pp_dwfwd(zp.data_ptr(), wp.data_ptr(), B, H, W, cin, cout, ss.data_ptr() if affine else None, relu,
                                 wd.data_ptr(), out.data_ptr(), st())
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert float(x3_untile(out, M, cout).abs().max()) > 0.1


@pytest.mark.parametrize("tile", [0, 3, 5, 6])
def test_gemm_batched_wgrad_form(L, tile):
    """Several weight-gradient problems of one shape in one launch == the same problems one by one."""
    rs = np.random.RandomState(11)
    nb, M, N, K = 5, 72, 136, 700
    Xs = [dev(rs.randn(K, M).astype(np.float32)) for _ in range(nb)]
    Ds = [dev(rs.randn(K, N).astype(np.float32)) for _ in range(nb)]
    Cs = [torch.full((M, N), float("nan"), device="cuda") for _ in range(nb)]
    a0, b0, c0 = Xs[0].data_ptr(), Ds[0].data_ptr(), Cs[0].data_ptr()
    table = torch.tensor([v for b in range(nb) for v in ((Xs[b].data_ptr() - a0) // 4, (Ds[b].data_ptr() - b0) // 4,
                                                         (Cs[b].data_ptr() - c0) // 4)], dtype=torch.int64, device="cuda")
    L.spnet_gemm_f32_batched(a0, b0, c0, table.data_ptr(), nb, 1, M, 1, N, N, M, N, K, tile, st())
    ws = torch.empty(WS, device="cuda")
    for b in range(nb):
        want = Xs[b].cpu().double().numpy().T @ Ds[b].cpu().double().numpy()
        close(Cs[b], want, rtol=2e-5, atol=3e-5 * np.sqrt(K))
        one = torch.empty(M, N, device="cuda")
        L.spnet_gemm_f32(Xs[b].data_ptr(), 1, M, Ds[b].data_ptr(), 1, N, one.data_ptr(), N, M, N, K, 1, ws.data_ptr(), WS,
                         None, tile if tile else 3, st())
        if tile:
            assert torch.equal(one, Cs[b])           # same tile, same k order -> same bits


@pytest.mark.parametrize("nb,M,N,K,ksplit", [(5, 72, 136, 700, 3), (10, 288, 32, 9744, 0), (20, 384, 1088, 2240, 0),
                                             (3, 96, 64, 130, 4), (4, 100, 36, 777, 2)])
def test_gemm_batched_wgrad_form_with_k_split(L, nb, M, N, K, ksplit):
    """Many same-shaped weight gradients in one launch with every K axis cut into slices (fp32 slabs + one batched slab
    reduce): against float64, and against the same problem run alone with the same tile and split (bit-identical below four slices).  ksplit 0 =
    the slice count the library suggests (spnet_gemm_batched_ksplit; Inception-ResNet-v2's block35 / block17 shapes)."""
    import ctypes
    rs = np.random.RandomState(nb + M)
    tile = ctypes.c_int(0)
    sugg = int(L.spnet_gemm_batched_ksplit(M, N, K, nb, ctypes.addressof(tile)))
    assert 1 <= sugg <= 64 and tile.value in (5, 6)
    ks = ksplit or sugg
    Xs = [dev(rs.randn(K, M).astype(np.float32)) for _ in range(nb)]
    Ds = [dev(rs.randn(K, N).astype(np.float32)) for _ in range(nb)]
    Cs = [torch.full((M, N), float("nan"), device="cuda") for _ in range(nb)]
    a0, b0, c0 = Xs[0].data_ptr(), Ds[0].data_ptr(), Cs[0].data_ptr()
    table = torch.tensor([v for b in range(nb) for v in ((Xs[b].data_ptr() - a0) // 4, (Ds[b].data_ptr() - b0) // 4,
                                                         (Cs[b].data_ptr() - c0) // 4)], dtype=torch.int64, device="cuda")
    ws = torch.full((nb * ks * M * N,), float("nan"), device="cuda")
    L.spnet_gemm_f32_batched_splitk(a0, b0, c0, table.data_ptr(), nb, 1, M, 1, N, N, M, N, K, tile.value, ks, ws.data_ptr(),
                                    ws.numel(), st())
    ws1 = torch.empty(max(ks, 1) * M * N, device="cuda")
    for b in (0, nb - 1):
        want = Xs[b].cpu().double().numpy().T @ Ds[b].cpu().double().numpy()
        close(Cs[b], want, rtol=2e-5, atol=3e-5 * np.sqrt(K))
        one = torch.empty(M, N, device="cuda")
        L.spnet_gemm_f32(Xs[b].data_ptr(), 1, M, Ds[b].data_ptr(), 1, N, one.data_ptr(), N, M, N, K, ks, ws1.data_ptr(),
                         ws1.numel(), None, tile.value, st())
        if ks < 4:
            assert torch.equal(one, Cs[b])           # same tile, same slices, same summation order -> same bits
        else:                                        # (the single-problem slab reduce adds >= 4 slabs in lane groups)
            close(Cs[b], one.cpu().double().numpy(), rtol=1e-5, atol=1e-5 * np.sqrt(K))
    if ks > 1:
        with pytest.raises(L.HipError):              # workspace too small for the slabs
            L.spnet_gemm_f32_batched_splitk(a0, b0, c0, table.data_ptr(), nb, 1, M, 1, N, N, M, N, K, tile.value, ks,
                                            ws.data_ptr(), nb * M * N, st())


def test_reduce_rows_with_scratch(L):
    """spnet_reduce_rows_ws: many rows folded through 32 parallel slices (a bias gradient over all pixels)."""
    rs = np.random.RandomState(5)
    for P, Lr in ((26320, 1088), (129, 20), (64, 320)):
        x = dev(rs.randn(P, Lr))
        out = torch.full((Lr,), float("nan"), device="cuda")
        scratch = torch.empty(32 * Lr, device="cuda")
        L.spnet_reduce_rows_ws(x.data_ptr(), P, Lr, out.data_ptr(), scratch.data_ptr(), scratch.numel(), st())
        close(out, x.cpu().double().numpy().sum(0), rtol=1e-5, atol=1e-5 * np.sqrt(P) * 4)
    with pytest.raises(L.HipError):
        L.spnet_reduce_rows_ws(x.data_ptr(), P, Lr, out.data_ptr(), scratch.data_ptr(), 10, st())


def test_reduce_rows_batched(L):
    """Several [P][L] -> [L] reductions in one launch, same bits as one spnet_reduce_rows call each."""
    rs = np.random.RandomState(4)
    shapes = [(32, 9 * 728), (2048, 9 * 64), (1, 40), (130, 9 * 1536), (17, 5)]
    ins = [dev(rs.randn(P, Lr).astype(np.float32)) for P, Lr in shapes]
    outs = [torch.full((Lr,), float("nan"), device="cuda") for _, Lr in shapes]
    jobs = torch.tensor([v for i, (P, Lr) in enumerate(shapes) for v in (ins[i].data_ptr(), outs[i].data_ptr(), P, Lr)],
                        dtype=torch.int64, device="cuda")
    L.spnet_reduce_rows_batched(jobs.data_ptr(), len(shapes), max(s[1] for s in shapes), st())
    for i, (P, Lr) in enumerate(shapes):
        close(outs[i], ins[i].cpu().double().numpy().sum(0), rtol=1e-5, atol=1e-4)
        if P <= 128:      # spnet_reduce_rows switches to two passes above 128 rows (different rounding order)
            one = torch.empty(Lr, device="cuda")
            L.spnet_reduce_rows(ins[i].data_ptr(), P, Lr, one.data_ptr(), st())
            assert torch.equal(one, outs[i])


def test_gemm_rejects_misaligned(L):
    a = torch.zeros(64, 6, device="cuda")
    with pytest.raises(L.HipError):
        L.spnet_gemm_f32(a.data_ptr(), 0, 6, a.data_ptr(), 1, 6, a.data_ptr(), 6, 8, 6, 6, 0, None, 0, None, 0, st())


class _DwForm:
    """The two implementations of the stride-1 depthwise layer behind one call shape: 'tiled' (LDS tiles) and
    'stream' / 'stream5' (row-marching waves; rows per wave chosen by the library / forced to 5, which cuts every test
    plane taller than 5 rows into segments with a ragged last one)."""

    def __init__(self, L, form):
        self.L, self.form = L, form
        self.rps = {"tiled": None, "stream": 0, "stream5": 5}[form]

    def rows(self, B, H, W, C):
        if self.rps is None:
            return self.L.spnet_dwconv3x3_tiled_rows(B, H, W, C)
        return self.L.spnet_dwconv3x3_stream_rows(B, H, W, C, self.rps)

    def ws(self, B, H, W, C):
        if self.rps is None:
            return self.L.spnet_dwconv3x3_tiled_bwd_ws(B, H, W, C)
        return self.L.spnet_dwconv3x3_stream_bwd_ws(B, H, W, C, self.rps)

    def fwd(self, x, w, y, B, H, W, C, relu_in, sc, sh):
        if self.rps is None:
            return self.L.spnet_dwconv3x3_tiled_fwd(x, w, y, B, H, W, C, relu_in, sc, sh, st())
        return self.L.spnet_dwconv3x3_stream_fwd(x, w, y, B, H, W, C, relu_in, sc, sh, self.rps, st())

    def bwd(self, dy, x, w, dx, dw, B, H, W, C, relu_in, add, ws, sc, sh, mu, inv, bnp, bnx):
        if self.rps is None:
            return self.L.spnet_dwconv3x3_tiled_bwd(dy, x, w, dx, dw, B, H, W, C, relu_in, add, ws, sc, sh, mu, inv, bnp, bnx, st())
        return self.L.spnet_dwconv3x3_stream_bwd(dy, x, w, dx, dw, B, H, W, C, relu_in, add, ws, sc, sh, mu, inv, bnp, bnx,
                                                 self.rps, st())


@pytest.mark.parametrize("B,H,W,C", [(2, 12, 16, 728), (3, 7, 5, 64), (1, 93, 125, 128), (2, 6, 8, 1536), (1, 1, 1, 8),
                                     (2, 24, 32, 256), (1, 47, 63, 128), (2, 13, 17, 40)])
@pytest.mark.parametrize("relu_in", [0, 1])
@pytest.mark.parametrize("fused_bn", [0, 1])
@pytest.mark.parametrize("form", ["tiled", "stream", "stream5"])
def test_dwconv_tiled(L, B, H, W, C, relu_in, fused_bn, form):
    """LDS-tiled forward and FUSED backward (data + weight gradient) -- the forms the engine uses.
    fused_bn=1: the input is a PRE-BatchNorm tensor whose affine is applied on load, and the backward
    also emits that BatchNorm's two backward sums."""
    rs = np.random.RandomState(C + H + 1)
    x = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32, requires_grad=True)
    w = torch.tensor(rs.randn(3, 3, C) * 0.3, dtype=torch.float32, requires_grad=True)
    add = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32)
    gamma = torch.tensor(rs.rand(C) + 0.5, dtype=torch.float32)
    beta = torch.tensor(rs.randn(C) * 0.3, dtype=torch.float32)
    if fused_bn:
        flat = x.reshape(-1, C)
        mu, var = flat.mean(0), flat.var(0, unbiased=False)
        invstd = torch.rsqrt(var + 1e-3)
        xhat = (x - mu) * invstd
        a = xhat * gamma + beta
        a.retain_grad()
        sc = (gamma * invstd).detach()
        sh = (beta - mu * gamma * invstd).detach()
    else:
        a = x
    y = T.dwconv3x3(torch.relu(a) if relu_in else a, w)
    dy = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32)
    (y * dy).sum().backward() if not fused_bn else ((y * dy).sum() + (a * add).sum()).backward()
    xd, wd, dyd, addd = x.detach().cuda(), w.detach().cuda(), dy.cuda(), add.cuda()
    scd = shd = mud = isd = None
    if fused_bn:
        scd, shd, mud, isd = sc.cuda(), sh.cuda(), mu.detach().cuda(), invstd.detach().cuda()
    P = lambda t: None if t is None else t.data_ptr()
    yd = torch.full_like(xd, float("nan"))
    F = _DwForm(L, form)
    F.fwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), B, H, W, C, relu_in, P(scd), P(shd))
    close(yd, y.detach(), rtol=1e-5, atol=2e-5)
    ws = torch.empty(F.ws(B, H, W, C), device="cuda")
    rows = F.rows(B, H, W, C)
    bnp = torch.full((rows, 2, C), float("nan"), device="cuda") if fused_bn else None
    dxd, dwd = torch.full_like(xd, float("nan")), torch.full((3, 3, C), float("nan"), device="cuda")
    F.bwd(dyd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), dwd.data_ptr(), B, H, W, C,
          relu_in, addd.data_ptr(), ws.data_ptr(), P(scd), P(shd), P(mud), P(isd), P(bnp), None)
    if fused_bn:
        # dx is the gradient wrt the BatchNorm OUTPUT a (incl. the added branch); the two sums are dbeta / dgamma/gamma-free
        close(dxd, a.grad, rtol=1e-5, atol=2e-5)
        sums = bnp.sum(0).cpu()
        close(sums[0], a.grad.reshape(-1, C).sum(0), rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W))
        close(sums[1], (a.grad * xhat.detach()).reshape(-1, C).sum(0), rtol=1e-4, atol=2e-4 * np.sqrt(B * H * W))
    else:
        close(dxd, x.grad + add, rtol=1e-5, atol=1e-5)
        F.bwd(dyd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), dwd.data_ptr(), B, H, W,
              C, relu_in, None, ws.data_ptr(), None, None, None, None, None, None)
        close(dxd, x.grad, rtol=1e-5, atol=1e-5)
    close(dwd, w.grad, rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W))
    if form != "tiled":      # y and dx of the streaming kernels are the tile kernels' bit for bit (same fmaf order)
        T_ = _DwForm(L, "tiled")
        y2, dx2 = torch.full_like(xd, float("nan")), torch.full_like(xd, float("nan"))
        T_.fwd(xd.data_ptr(), wd.data_ptr(), y2.data_ptr(), B, H, W, C, relu_in, P(scd), P(shd))
        assert torch.equal(y2, yd)
        ws2 = torch.empty(T_.ws(B, H, W, C), device="cuda")
        bnp2 = torch.empty((T_.rows(B, H, W, C), 2, C), device="cuda") if fused_bn else None
        T_.bwd(dyd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dx2.data_ptr(), dwd.data_ptr(), B, H, W, C, relu_in,
               addd.data_ptr() if fused_bn else None, ws2.data_ptr(), P(scd), P(shd), P(mud), P(isd), P(bnp2), None)
        assert torch.equal(dx2, dxd)


@pytest.mark.parametrize("B,H,W,C", [(2, 12, 16, 728), (2, 6, 8, 1536), (1, 13, 17, 40)])
@pytest.mark.parametrize("form", ["tiled", "stream", "stream5"])
def test_dwconv_tiled_bwd_sums_for_a_block_output(L, B, H, W, C, form):
    """bn_x: the layer reads a block OUTPUT y = BN(yp) + residual (ReLU on load, no affine) and still emits the
    backward sums of that BatchNorm, taking xhat from yp."""
    rs = np.random.RandomState(C + W)
    yp = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32)
    res = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32)
    w = torch.tensor(rs.randn(3, 3, C) * 0.3, dtype=torch.float32)
    add = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32)
    gamma, beta = torch.tensor(rs.rand(C) + 0.5, dtype=torch.float32), torch.tensor(rs.randn(C) * 0.3, dtype=torch.float32)
    flat = yp.reshape(-1, C)
    mu, invstd = flat.mean(0), torch.rsqrt(flat.var(0, unbiased=False) + 1e-3)
    xhat = (yp - mu) * invstd
    y = (xhat * gamma + beta + res).requires_grad_(True)
    z = T.dwconv3x3(torch.relu(y), w)
    dz = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32)
    ((z * dz).sum() + (y * add).sum()).backward()
    g = y.grad                                          # gradient wrt the block output = wrt the BN output
    yd, ypd, wd, dzd, addd = y.detach().cuda(), yp.cuda(), w.cuda(), dz.cuda(), add.cuda()
    F = _DwForm(L, form)
    ws = torch.empty(F.ws(B, H, W, C), device="cuda")
    rows = F.rows(B, H, W, C)
    bnp = torch.full((rows, 2, C), float("nan"), device="cuda")
    dxd, dwd = torch.full_like(yd, float("nan")), torch.full((3, 3, C), float("nan"), device="cuda")
    mud, isd = mu.cuda(), invstd.cuda()
    F.bwd(dzd.data_ptr(), yd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), dwd.data_ptr(), B, H, W, C, 1,
          addd.data_ptr(), ws.data_ptr(), None, None, mud.data_ptr(), isd.data_ptr(), bnp.data_ptr(), ypd.data_ptr())
    close(dxd, g, rtol=1e-5, atol=2e-5)
    sums = bnp.sum(0).cpu()
    close(sums[0], g.reshape(-1, C).sum(0), rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W))
    close(sums[1], (g * xhat).reshape(-1, C).sum(0), rtol=1e-4, atol=2e-4 * np.sqrt(B * H * W))
    with pytest.raises(L.HipError):                     # bn_x without bn_partial makes no sense
        F.bwd(dzd.data_ptr(), yd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), dwd.data_ptr(), B, H, W, C,
              1, None, ws.data_ptr(), None, None, None, None, None, ypd.data_ptr())


@pytest.mark.parametrize("M,C,P,act,res", [(6144, 728, 64, 0, 1), (1536, 2048, 12, 1, 0), (700, 36, 128, 0, 1), (300, 260, 3, 2, 0)])
def test_batchnorm_finalize_folded_into_the_apply_pass(L, M, C, P, act, res):
    """spnet_bn_finalize_apply == spnet_bn_finalize_fwd + spnet_bn_apply, bit for bit (one launch while P <= 128)."""
    rs = np.random.RandomState(M + C)
    x, r = dev(rs.randn(M, C)), dev(rs.randn(M, C))
    part = dev(rs.randn(P, 2, C) * 3)
    part[:, 1] = part[:, 1].abs() * 40 + 10
    gamma, beta = dev(rs.rand(C) + 0.5), dev(rs.randn(C) * 0.3)
    outs = []
    for fused in (0, 1):
        mm, mv = dev(np.full(C, 0.25)), dev(np.full(C, 0.75))
        save, ss = torch.full((2 * C,), float("nan"), device="cuda"), torch.full((2 * C,), float("nan"), device="cuda")
        y = torch.full((M, C), float("nan"), device="cuda")
        if fused:
            L.spnet_bn_finalize_apply(part.data_ptr(), P, x.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(),
                                      mv.data_ptr(), save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), act,
                                      r.data_ptr() if res else None, y.data_ptr(), 1e-3, 0.99, st())
        else:
            L.spnet_bn_finalize_fwd(part.data_ptr(), P, M, C, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                    save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), 1e-3, 0.99, st())
            L.spnet_bn_apply(x.data_ptr(), M, C, ss.data_ptr(), act, r.data_ptr() if res else None, 0, y.data_ptr(), st())
        torch.cuda.synchronize()
        outs.append((y, save, ss, mm, mv))
    for a, b in zip(*outs):
        assert not torch.isnan(b).any()
        assert torch.equal(a, b)


@pytest.mark.parametrize("P,C", [(11625, 128), (1024, 64), (2961, 256), (1031, 36), (1500, 728)])
def test_batchnorm_finalize_from_many_partial_rows(L, P, C):
    """>= 1,024 partial rows (the entry flow's GEMM epilogues leave up to 11,625) take the two-stage finalize: slice sums
    left in place as (hi, lo) float pairs, then combined.  Against the float64 sums of the same rows; the partial buffer
    is the caller's scratch and may be overwritten.  A second run from a fresh copy gives the same bits."""
    rs = np.random.RandomState(P + C)
    M = 32 * P
    part_h = (rs.randn(P, 2, C) * 3).astype(np.float32)
    part_h[:, 1] = np.abs(part_h[:, 1]) * 40 + 10
    s, q = part_h[:, 0].astype(np.float64).sum(0), part_h[:, 1].astype(np.float64).sum(0)
    mu = s / M
    var = np.maximum(q / M - mu * mu, 0.0)
    gamma, beta = dev(rs.rand(C) + 0.5), dev(rs.randn(C) * 0.3)
    runs = []
    for _ in range(2):
        part = dev(part_h)
        mm, mv = dev(np.full(C, 0.25)), dev(np.full(C, 0.75))
        save, ss = torch.full((2 * C,), float("nan"), device="cuda"), torch.full((2 * C,), float("nan"), device="cuda")
        L.spnet_bn_finalize_fwd(part.data_ptr(), P, M, C, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), 1e-3, 0.99, st())
        torch.cuda.synchronize()
        runs.append((save, ss, mm, mv))
    save, ss, mm, mv = runs[0]
    close(save[:C], mu, rtol=2e-7, atol=1e-7 * np.abs(mu).max())
    close(save[C:], 1 / np.sqrt(var + 1e-3), rtol=2e-7, atol=0)
    sc = gamma.cpu().double().numpy() / np.sqrt(var + 1e-3)
    close(ss[:C], sc, rtol=3e-7, atol=0)
    close(ss[C:], beta.cpu().double().numpy() - mu * sc, rtol=1e-6, atol=1e-6)
    close(mm, 0.99 * 0.25 + 0.01 * mu, rtol=1e-6, atol=1e-7)
    close(mv, 0.99 * 0.75 + 0.01 * var * M / (M - 1), rtol=1e-6, atol=1e-7)
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,C,P", [(9744, 32, 153), (2240, 192, 24), (384, 256, 4), (9744, 96, 300)])
def test_batchnorm_forward_into_a_column_block(L, M, C, P):
    """The `_ld` forms (finalize + apply, stand-alone training forward, inference forward) with the output a column block
    of a wider tensor == the dense forms, bit for bit, and nothing outside the block is touched."""
    rs = np.random.RandomState(M + C)
    x = dev(rs.randn(M, C))
    part = dev(rs.randn(P, 2, C) * 3)
    part[:, 1] = part[:, 1].abs() * 40 + 10
    gamma, beta = dev(rs.rand(C) + 0.5), dev(rs.randn(C) * 0.3)
    wide, off = C + 64, 32

    def fresh():
        return (dev(np.full(C, 0.25)), dev(np.full(C, 0.75)), torch.full((2 * C,), float("nan"), device="cuda"),
                torch.full((2 * C,), float("nan"), device="cuda"))

    ws = torch.empty(L.spnet_bn_ws(M, C) + 16, device="cuda")
    for kind in ("finalize_apply", "train", "infer"):
        outs = []
        for ld in (C, wide):
            mm, mv, save, ss = fresh()
            buf = torch.full((M, ld), -7.0, device="cuda")
            y = buf.data_ptr() + (4 * off if ld != C else 0)
            if kind == "finalize_apply":
                L.spnet_bn_finalize_apply_ld(part.clone().data_ptr(), P, x.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(),
                                             mm.data_ptr(), mv.data_ptr(), save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), 1,
                                             None, y, ld, 1e-3, 0.99, st())
            elif kind == "train":
                L.spnet_bn_fwd_train_ld(x.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                        save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), 1, None, 0, y, ld, 1e-3, 0.99,
                                        ws.data_ptr(), st())
            else:
                L.spnet_bn_fwd_infer_ld(x.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                        ss.data_ptr(), 1, None, 0, y, ld, 1e-3, st())
            torch.cuda.synchronize()
            outs.append((buf, ss))
        dense, blocked = outs[0][0], outs[1][0]
        assert torch.equal(blocked[:, off:off + C], dense) and torch.equal(outs[0][1], outs[1][1])
        assert float((blocked[:, :off] + 7.0).abs().max()) == 0.0 and float((blocked[:, off + C:] + 7.0).abs().max()) == 0.0
    with pytest.raises(L.HipError):                     # stride narrower than the block
        L.spnet_bn_fwd_infer_ld(x.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                ss.data_ptr(), 1, None, 0, buf.data_ptr(), C - 4, 1e-3, st())


@pytest.mark.parametrize("M,C", [(6144, 728), (1536, 1536), (500, 36)])
def test_batchnorm_backward_from_partial_sums(L, M, C):
    """spnet_bn_bwd_from_partials against the closed form (float64), through BOTH of its paths: the one-launch form
    (<= 128 partial rows: finalize folded into the apply pass) and finalize + apply (more rows); the two agree bit for
    bit when fed the same sums."""
    rs = np.random.RandomState(M)
    x, g = rs.randn(M, C).astype(np.float32), rs.randn(M, C).astype(np.float32)
    gamma = (rs.rand(C) + 0.5).astype(np.float32)
    mu, var = x.astype(np.float64).mean(0), x.astype(np.float64).var(0)
    invstd = 1.0 / np.sqrt(var + 1e-3)
    xh = (x - mu) * invstd
    sg, sgx = g.astype(np.float64).sum(0), (g * xh).sum(0)
    want = gamma * invstd * (g - sg / M - xh * sgx / M)
    xd, gd, gam = dev(x), dev(g), dev(gamma)
    mud, isd = dev(mu), dev(invstd)
    res = []
    for P in (32, 160):
        part = np.zeros((P, 2, C), np.float32)          # the same sums, spread over the first two rows
        part[0, 0], part[0, 1] = sg, sgx
        dx = torch.full((M, C), float("nan"), device="cuda")
        dga, dbe = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        co = torch.empty(3 * C, device="cuda")
        L.spnet_bn_bwd_from_partials(xd.data_ptr(), gd.data_ptr(), M, C, gam.data_ptr(), gam.data_ptr(), mud.data_ptr(),
                                     isd.data_ptr(), P, dev(part).data_ptr(), dx.data_ptr(), dga.data_ptr(), dbe.data_ptr(),
                                     co.data_ptr(), st())
        close(dx, want, rtol=1e-4, atol=1e-4)
        close(dbe, sg, rtol=1e-6, atol=1e-4)
        close(dga, sgx, rtol=1e-6, atol=1e-4)
        res.append((dx, dga, dbe))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,W,C,rows,relu_in", [(4, 12, 16, 728, 64, 1), (2, 6, 8, 1536, 12, 1), (3, 24, 32, 128, 128, 0),
                                                  (2, 12, 16, 260, 7, 1)])
def test_dwconv_tiled_forward_with_the_producer_batchnorm_finalize_folded_in(L, B, H, W, C, rows, relu_in):
    """spnet_dwconv3x3_tiled_fwd_bnfin == spnet_bn_finalize_fwd followed by spnet_dwconv3x3_tiled_fwd, bit for bit:
    output, saved statistics, scale/shift and the moving-statistics update (the engine's middle / exit flow)."""
    rs = np.random.RandomState(C + rows)
    M = B * H * W
    x = dev(rs.randn(B, H, W, C))
    w = dev(rs.randn(3, 3, C) * 0.3)
    part = dev(rs.randn(rows, 2, C) * 3)
    part[:, 1] = part[:, 1].abs() * 40 + 10          # sums of squares
    gamma, beta = dev(rs.rand(C) + 0.5), dev(rs.randn(C) * 0.3)

    def fresh():
        return (dev(rs.randn(C) * 0 + 0.25), dev(np.ones(C) * 0.75), torch.full((2 * C,), float("nan"), device="cuda"),
                torch.full((2 * C,), float("nan"), device="cuda"), torch.full((B, H, W, C), float("nan"), device="cuda"))

    mm0, mv0, save0, ss0, y0 = fresh()
    L.spnet_bn_finalize_fwd(part.data_ptr(), rows, M, C, gamma.data_ptr(), beta.data_ptr(), mm0.data_ptr(), mv0.data_ptr(),
                            save0.data_ptr(), save0[C:].data_ptr(), ss0.data_ptr(), 1e-3, 0.99, st())
    L.spnet_dwconv3x3_tiled_fwd(x.data_ptr(), w.data_ptr(), y0.data_ptr(), B, H, W, C, relu_in, ss0.data_ptr(),
                                ss0[C:].data_ptr(), st())
    mm1, mv1, save1, ss1, y1 = fresh()
    L.spnet_dwconv3x3_tiled_fwd_bnfin(x.data_ptr(), w.data_ptr(), y1.data_ptr(), B, H, W, C, relu_in, part.data_ptr(), rows, M,
                                      gamma.data_ptr(), beta.data_ptr(), mm1.data_ptr(), mv1.data_ptr(), save1.data_ptr(),
                                      save1[C:].data_ptr(), ss1.data_ptr(), 1e-3, 0.99, st())
    torch.cuda.synchronize()
    for a, b in ((y0, y1), (save0, save1), (ss0, ss1), (mm0, mm1), (mv0, mv1)):
        assert not torch.isnan(b).any()
        assert torch.equal(a, b)
    with pytest.raises(L.HipError):                     # more partial rows than the prologue is written for
        L.spnet_dwconv3x3_tiled_fwd_bnfin(x.data_ptr(), w.data_ptr(), y1.data_ptr(), B, H, W, C, relu_in, part.data_ptr(), 129,
                                          M, gamma.data_ptr(), beta.data_ptr(), mm1.data_ptr(), mv1.data_ptr(),
                                          save1.data_ptr(), save1[C:].data_ptr(), ss1.data_ptr(), 1e-3, 0.99, st())


@pytest.mark.parametrize("M,N,K", [(6144, 728, 728), (372, 128, 64), (1536, 2048, 1536), (100, 64, 288), (33, 72, 40)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5, 6, 7, 8, 9, 10])
def test_gemm_colstats_and_bn_finalize(L, M, N, K, tile):
    import ctypes
    rs = np.random.RandomState(M + N)
    A, Bm = rs.randn(M, K).astype(np.float32), (rs.randn(K, N) * 0.1).astype(np.float32)
    a, b = dev(A), dev(Bm)
    c = torch.empty(M, N, device="cuda")
    part = torch.full(((M + 31) // 32 * 2 * N,), float("nan"), device="cuda")
    rows = ctypes.c_int(0)
    L.spnet_gemm_f32_colstats(a.data_ptr(), 0, K, b.data_ptr(), 1, N, c.data_ptr(), N, M, N, K, tile, part.data_ptr(),
                              ctypes.addressof(rows), st())
    Cref = A.astype(np.float64) @ Bm.astype(np.float64)
    close(c, Cref, rtol=2e-5, atol=2e-5 * np.sqrt(K))
    p = part[:rows.value * 2 * N].reshape(rows.value, 2, N).sum(0).cpu()
    close(p[0], Cref.sum(0), rtol=1e-4, atol=1e-3 * np.sqrt(M))
    close(p[1], (Cref ** 2).sum(0), rtol=1e-4, atol=1e-3)
    # finalize from those partials == oracle batch statistics
    gamma, beta = dev(rs.rand(N) + 0.5), dev(rs.randn(N))
    mm, mv = dev(np.zeros(N)), dev(np.ones(N))
    save, ss = torch.empty(2 * N, device="cuda"), torch.empty(2 * N, device="cuda")
    L.spnet_bn_finalize_fwd(part.data_ptr(), rows.value, M, N, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                            save.data_ptr(), save[N:].data_ptr(), ss.data_ptr(), 1e-3, 0.99, st())
    mu, var = Cref.mean(0), Cref.var(0)
    close(save[:N], mu, rtol=1e-4, atol=1e-5)
    close(save[N:], 1 / np.sqrt(var + 1e-3), rtol=1e-4, atol=1e-5)
    sc = gamma.cpu().double().numpy() / np.sqrt(var + 1e-3)
    close(ss[:N], sc, rtol=1e-4, atol=1e-5)
    close(ss[N:], beta.cpu().double().numpy() - mu * sc, rtol=1e-4, atol=1e-4)
    y = torch.empty_like(c)
    L.spnet_bn_apply(c.data_ptr(), M, N, ss.data_ptr(), 1, None, 0, y.data_ptr(), st())
    close(y, np.maximum((Cref - mu) * sc + beta.cpu().double().numpy(), 0), rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("M,C", [(6144, 728), (50, 64), (23250, 128), (4096, 3), (7, 2048), (3000, 32)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_batchnorm_train_and_backward(L, M, C, act):
    rs = np.random.RandomState(M + C + act)
    x = torch.tensor(rs.randn(M, C) * 1.5 + 0.3, dtype=torch.float32, requires_grad=True)
    gamma = torch.tensor(rs.rand(C) + 0.5, dtype=torch.float32, requires_grad=True)
    beta = torch.tensor(rs.randn(C) * 0.2, dtype=torch.float32, requires_grad=True)
    mm, mv = torch.tensor(rs.randn(C), dtype=torch.float32), torch.tensor(rs.rand(C) + 0.5, dtype=torch.float32)
    res = torch.tensor(rs.randn(M, C), dtype=torch.float32)
    mm_o, mv_o = mm.clone(), mv.clone()
    yb = T.batchnorm(x, gamma, beta, mm_o, mv_o, training=True)
    y = {0: lambda t: t, 1: torch.relu, 2: lambda t: torch.nn.functional.leaky_relu(t, 0.1)}[act](yb) + res
    dy = torch.tensor(rs.randn(M, C), dtype=torch.float32)
    y.backward(dy)

    xd, gd, bd, mmd, mvd, resd, dyd = (t.detach().cuda() for t in (x, gamma, beta, mm, mv, res, dy))
    save = torch.empty(2 * C + 8, device="cuda")
    ss = torch.empty(2 * C + 8, device="cuda")
    ws = torch.empty(L.spnet_bn_ws(M, C), device="cuda")
    yd = torch.empty_like(xd)
    L.spnet_bn_fwd_train(xd.data_ptr(), M, C, gd.data_ptr(), bd.data_ptr(), mmd.data_ptr(), mvd.data_ptr(),
                         save.data_ptr(), save[C:].data_ptr(), ss.data_ptr(), act, resd.data_ptr(), 0, yd.data_ptr(),
                         1e-3, 0.99, ws.data_ptr(), st())
    close(yd, y.detach(), rtol=2e-5, atol=2e-5)
    close(mmd, mm_o, rtol=1e-5, atol=1e-6)
    close(mvd, mv_o, rtol=1e-5, atol=1e-6)
    co = torch.empty(3 * C + 8, device="cuda")
    dxd, dgd, dbd = torch.empty_like(xd), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    L.spnet_bn_bwd(xd.data_ptr(), dyd.data_ptr(), M, C, gd.data_ptr(), bd.data_ptr(), save.data_ptr(),
                   save[C:].data_ptr(), act, dxd.data_ptr(), dgd.data_ptr(), dbd.data_ptr(), co.data_ptr(),
                   ws.data_ptr(), st())
    close(dxd, x.grad, rtol=1e-4, atol=2e-5)
    close(dgd, gamma.grad, rtol=1e-4, atol=1e-4 * np.sqrt(M))
    close(dbd, beta.grad, rtol=1e-4, atol=1e-4 * np.sqrt(M))
    # inference form
    yi = T.batchnorm(x.detach(), gamma.detach(), beta.detach(), mm, mv, training=False)
    mm2, mv2 = dev(mm), dev(mv)          # keep the device copies alive across the launch
    L.spnet_bn_fwd_infer(xd.data_ptr(), M, C, gd.data_ptr(), bd.data_ptr(), mm2.data_ptr(), mv2.data_ptr(),
                         ss.data_ptr(), 0, None, 0, yd.data_ptr(), 1e-3, st())
    close(yd, yi, rtol=2e-5, atol=2e-5)


def test_batchnorm_broadcast_residual(L):
    rs = np.random.RandomState(0)
    M, C = 1000, 3
    x, res = dev(rs.randn(M, C)), dev(rs.randn(M))
    g, b = dev(np.ones(C)), dev(np.zeros(C))
    mm, mv = dev(np.zeros(C)), dev(np.ones(C))
    ss = torch.empty(16, device="cuda")
    y = torch.empty_like(x)
    L.spnet_bn_fwd_infer(x.data_ptr(), M, C, g.data_ptr(), b.data_ptr(), mm.data_ptr(), mv.data_ptr(), ss.data_ptr(),
                         0, res.data_ptr(), 1, y.data_ptr(), 1e-3, st())
    want = x.cpu() / np.sqrt(1 + 1e-3) + res.cpu()[:, None]
    close(y, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,H,W,C", [(2, 12, 16, 728), (1, 93, 125, 128), (2, 5, 5, 64), (1, 4, 4, 8), (2, 47, 63, 256)])
def test_maxpool_add(L, B, H, W, C):
    rs = np.random.RandomState(H * W)
    x = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32, requires_grad=True)
    OH, OW = (H + 1) // 2, (W + 1) // 2
    res = torch.tensor(rs.randn(B, OH, OW, C), dtype=torch.float32)
    y = T.maxpool3x3s2_same(x) + res
    dy = torch.tensor(rs.randn(B, OH, OW, C), dtype=torch.float32)
    y.backward(dy)
    xd = x.detach().cuda()
    yd = torch.empty(B, OH, OW, C, device="cuda")
    idx = torch.empty(B * OH * OW * C // 4, dtype=torch.int32, device="cuda")
    resd, dyd = res.cuda(), dy.cuda()
    L.spnet_maxpool3x3s2_add_fwd(xd.data_ptr(), resd.data_ptr(), yd.data_ptr(), idx.data_ptr(), B, H, W, C, None, None, st())
    assert torch.equal(yd.cpu(), y.detach())          # max + one add: bit-exact
    # with both BatchNorm affines applied on load
    xs, rss = dev(np.concatenate([rs.randn(C), rs.randn(C)])), dev(np.concatenate([rs.randn(C), rs.randn(C)]))
    ya = T.maxpool3x3s2_same(x.detach() * xs[:C].cpu() + xs[C:].cpu()) + (res * rss[:C].cpu() + rss[C:].cpu())
    yd2 = torch.empty_like(yd)
    L.spnet_maxpool3x3s2_add_fwd(xd.data_ptr(), resd.data_ptr(), yd2.data_ptr(), None, B, H, W, C, xs.data_ptr(), rss.data_ptr(), st())
    close(yd2, ya, rtol=1e-5, atol=1e-5)
    dxd = torch.empty_like(xd)
    L.spnet_maxpool3x3s2_bwd(dyd.data_ptr(), idx.data_ptr(), dxd.data_ptr(), B, H, W, C, st())
    close(dxd, x.grad, rtol=1e-6, atol=1e-6)
    # the same gradient with the BatchNorm-backward sums of the pooled layer taken in the same pass
    rows = L.spnet_maxpool3x3s2_bwd_rows(B, H, W, C, 128 if C != 128 else 1024)
    assert 1 <= rows <= 1024
    yp = dev(rs.randn(B, H, W, C))
    mu, istd = dev(rs.randn(C) * 0.2), dev(rs.rand(C) + 0.5)
    part = torch.full((rows, 2, C), float("nan"), device="cuda")
    dx2 = torch.full_like(xd, float("nan"))
    L.spnet_maxpool3x3s2_bwd_bnsums(dyd.data_ptr(), idx.data_ptr(), dx2.data_ptr(), B, H, W, C, yp.data_ptr(), mu.data_ptr(),
                                    istd.data_ptr(), part.data_ptr(), rows, st())
    assert torch.equal(dx2, dxd)
    g64 = dxd.cpu().double().reshape(-1, C)
    xh = (yp.cpu().double().reshape(-1, C) - mu.cpu().double()) * istd.cpu().double()
    sums = part.sum(0).cpu()
    close(sums[0], g64.sum(0), rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W))
    close(sums[1], (g64 * xh).sum(0), rtol=1e-4, atol=2e-4 * np.sqrt(B * H * W))


@pytest.mark.parametrize("C", [1, 3])
def test_avgpool(L, C):
    rs = np.random.RandomState(C)
    B, H, W = 2, 9, 14
    x = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32, requires_grad=True)
    y = T.avgpool2(x)
    dy = torch.tensor(rs.randn(*y.shape), dtype=torch.float32)
    y.backward(dy)
    yd = torch.empty(*y.shape, device="cuda")
    xd, dyd = x.detach().cuda(), dy.cuda()
    L.spnet_avgpool2_fwd(xd.data_ptr(), yd.data_ptr(), B, H, W, C, st())
    close(yd, y.detach(), rtol=1e-6, atol=1e-6)
    dxd = torch.empty(B, H, W, C, device="cuda")
    L.spnet_avgpool2_bwd(dyd.data_ptr(), dxd.data_ptr(), B, H, W, C, st())
    close(dxd, x.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("cin,cout,stride,same,H,W", [(1, 3, 1, 1, 20, 28), (3, 3, 1, 1, 20, 28), (3, 32, 2, 0, 21, 28),
                                                      (3, 32, 2, 0, 48, 64), (1, 3, 1, 1, 96, 128),
                                                      # widths that are multiples of 8: the 3 -> 3 weight gradient by pixel runs
                                                      (3, 3, 1, 1, 7, 8), (3, 3, 1, 1, 12, 32), (3, 3, 1, 1, 48, 64),
                                                      (3, 3, 1, 1, 1, 16), (3, 3, 1, 1, 5, 4), (3, 3, 1, 1, 192, 256)])
def test_small_conv(L, cin, cout, stride, same, H, W):
    rs = np.random.RandomState(cin * cout + H)
    B = 2
    x = torch.tensor(rs.randn(B, H, W, cin), dtype=torch.float32, requires_grad=True)
    w = torch.tensor(rs.randn(3, 3, cin, cout) * 0.3, dtype=torch.float32, requires_grad=True)
    y = T.conv2d(x, w, stride, "same" if same else "valid")
    dy = torch.tensor(rs.randn(*y.shape), dtype=torch.float32)
    y.backward(dy)
    ws = torch.empty(WS, device="cuda")
    xd, wd, dyd = x.detach().cuda(), w.detach().cuda(), dy.cuda()
    yd = torch.empty(*y.shape, device="cuda")
    L.spnet_conv3x3_small(0, cin, cout, stride, same, xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), B, H, W, ws.data_ptr(), WS, st())
    close(yd, y.detach(), rtol=1e-5, atol=1e-5)
    dxd = torch.empty_like(xd)
    L.spnet_conv3x3_small(1, cin, cout, stride, same, dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), B, H, W, ws.data_ptr(), WS, st())
    close(dxd, x.grad, rtol=1e-5, atol=2e-5)
    dwd = torch.empty_like(wd)
    L.spnet_conv3x3_small(2, cin, cout, stride, same, xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), B, H, W, ws.data_ptr(), WS, st())
    close(dwd, w.grad, rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W))
    if cin == 3 and cout == 3 and W % 4 == 0:
        # round 5: the runs-of-four-pixels kernels (aligned buffers) against the one-thread-per-pixel kernels the library
        # falls back to when a buffer is only 4-byte aligned: the same fmaf chains, bit for bit
        def off(t):
            buf = torch.empty(t.numel() + 4, device="cuda")
            v = buf[1:1 + t.numel()].view(t.shape)
            v.copy_(t)
            return buf, v
        (_, xo), (_, dyo) = off(xd), off(dyd)
        (_, yo), (_, dxo) = off(yd), off(dxd)
        assert xo.data_ptr() % 16 == 4
        L.spnet_conv3x3_small(0, cin, cout, stride, same, xo.data_ptr(), wd.data_ptr(), yo.data_ptr(), B, H, W, ws.data_ptr(), WS, st())
        L.spnet_conv3x3_small(1, cin, cout, stride, same, dyo.data_ptr(), wd.data_ptr(), dxo.data_ptr(), B, H, W, ws.data_ptr(), WS, st())
        assert torch.equal(yo, yd) and torch.equal(dxo, dxd)


@pytest.mark.parametrize("B,H,W", [(2, 9, 40), (1, 33, 47), (3, 5, 34), (2, 40, 70)])
def test_conv3x3_implicit_gemm(L, B, H, W):
    """block1_conv2 without the im2col matrix: forward, input gradient and weight gradient against the oracle
    conv and its autograd."""
    rs = np.random.RandomState(H * W)
    x = torch.tensor(rs.randn(B, H, W, 32), dtype=torch.float32, requires_grad=True)
    w = torch.tensor(rs.randn(3, 3, 32, 64) * 0.1, dtype=torch.float32, requires_grad=True)
    y = T.conv2d(x, w, 1, "valid")
    dy = torch.tensor(rs.randn(*y.shape), dtype=torch.float32)
    y.backward(dy)
    xd, dyd, wd = x.detach().cuda(), dy.cuda(), w.detach().cuda()
    yd = torch.full(tuple(y.shape), float("nan"), device="cuda")
    L.spnet_conv3x3_fwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), B, H, W, 32, 64, st())
    close(yd, y.detach(), rtol=1e-4, atol=1e-4)
    dxd = torch.full((B, H, W, 32), float("nan"), device="cuda")
    L.spnet_conv3x3_dgrad(dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), B, H, W, 32, 64, st())
    close(dxd, x.grad, rtol=1e-4, atol=1e-4)
    n = L.spnet_conv3x3_wgrad_ws(B, H, W, 32, 64)
    ws = torch.empty(n, device="cuda")
    dwd = torch.full((3, 3, 32, 64), float("nan"), device="cuda")
    L.spnet_conv3x3_wgrad(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), B, H, W, 32, 64, ws.data_ptr(), n, st())
    close(dwd, w.grad, rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W))
    dw2 = torch.empty_like(dwd)
    L.spnet_conv3x3_wgrad(xd.data_ptr(), dyd.data_ptr(), dw2.data_ptr(), B, H, W, 32, 64, ws.data_ptr(), n, st())
    assert torch.equal(dwd, dw2)                     # fixed reduction order
    with pytest.raises(L.HipError):
        L.spnet_conv3x3_dgrad(dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), B, H, W, 16, 64, st())
    with pytest.raises(L.HipError):
        L.spnet_conv3x3_wgrad(xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), B, H, W, 32, 64, ws.data_ptr(), n - 1, st())


def test_gather_scatter_stride2(L):
    rs = np.random.RandomState(9)
    B, C = 2, 32
    for (h, w_) in [(9, 11), (8, 6)]:
        xx = dev(rs.randn(B, h, w_, C))
        xs = torch.empty(B, (h + 1) // 2, (w_ + 1) // 2, C, device="cuda")
        L.spnet_gather_s2(xx.data_ptr(), xs.data_ptr(), B, h, w_, C, st())
        assert torch.equal(xs.cpu(), xx.cpu()[:, ::2, ::2, :])
        acc = dev(rs.randn(B, h, w_, C))
        want = acc.cpu().clone()
        want[:, ::2, ::2, :] += xs.cpu()
        L.spnet_scatter_add_s2(xs.data_ptr(), acc.data_ptr(), B, h, w_, C, st())
        assert torch.equal(acc.cpu(), want)


@pytest.mark.parametrize("loss_type", ["same", "hybrid"])
def test_ellipse_loss(L, golden, loss_type):
    for tag in ("loss", "loss2"):
        yt, yp = golden[f"{tag}_yt"], golden[f"{tag}_yp"]
        B = yt.shape[0]
        g = torch.empty(B, 576, device="cuda")
        parts, out = torch.empty(B, 5, device="cuda"), torch.empty(6, device="cuda")
        ytd, ypd = dev(yt), dev(yp)
        L.spnet_ellipse_loss(ytd.data_ptr(), ypd.data_ptr(), g.data_ptr(), parts.data_ptr(), out.data_ptr(), B, 576,
                             0 if loss_type == "same" else 1, st())
        # against the reference's own my_loss output (golden) ...
        close(out[:5], golden[f"{tag}_{loss_type}_parts"], rtol=1e-5, atol=1e-9)
        close(out[5], golden[f"{tag}_{loss_type}_total"], rtol=1e-5, atol=1e-9)
        # ... and the closed-form gradient of the oracle
        close(g, R.loss_grad(yt, yp, loss_type), rtol=1e-5, atol=1e-9)


def test_decode(L):
    rs = np.random.RandomState(4)
    Yn = (rs.randn(5, 576) * 0.5).astype(np.float32)
    gc = R.grid_constants()
    ynd, md, rd = dev(Yn), dev(gc["means"]), dev(gc["ranges"])
    for sig in (0, 1):
        out = torch.empty(5, 72, 7, device="cuda")
        L.spnet_decode(ynd.data_ptr(), md.data_ptr(), rd.data_ptr(), out.data_ptr(), 5, 576, sig, st())
        want = R.decode(Yn, "hybrid" if sig else "same")
        # angle: atan2 in f32 on both sides, allow a few ulp of 180 degrees
        close(out, want, rtol=1e-5, atol=2e-4)


def test_adam_matches_keras_form(L):
    rs = np.random.RandomState(5)
    n, l2n = 4096 + 64, 1024
    p, g = rs.randn(n).astype(np.float32), (rs.randn(n) * 0.1).astype(np.float32)
    m, v = (rs.randn(n) * 0.01).astype(np.float32), (rs.rand(n) * 0.01).astype(np.float32)
    pd, gd, md, vd = dev(p), dev(g), dev(m), dev(v)
    sq, l2out = torch.empty(2048, device="cuda"), torch.empty(2, device="cuda")
    t, lr = 7, 3e-4
    import math
    lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    L.spnet_adam_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, l2n, lr_t, 0.9, 0.999, 1e-7, 1e-4, 0.5,
                      None, sq.data_ptr(), l2out.data_ptr(), None, st())
    wp, wm, wv = p.copy(), m.copy(), v.copy()
    wp[:l2n], wm[:l2n], wv[:l2n] = R.adam_step(p[:l2n], g[:l2n] * 0.5, m[:l2n], v[:l2n], t, lr, l2=1e-4)
    wp[l2n:], wm[l2n:], wv[l2n:] = R.adam_step(p[l2n:], g[l2n:] * 0.5, m[l2n:], v[l2n:], t, lr)
    close(pd, wp, rtol=1e-6, atol=1e-7)
    close(md, wm, rtol=1e-6, atol=1e-8)
    close(vd, wv, rtol=1e-6, atol=1e-9)
    close(l2out[0], 1e-4 * np.sum(p[:l2n].astype(np.float64) ** 2), rtol=1e-5, atol=0)


def test_adam_in_two_ranges_equals_the_one_launch_step(L):
    """spnet_adam_part over [0, cut) and [cut, n) + spnet_adam_l2_sum (the engine's optimizer step: the Dense head's range
    early, the rest at the end of backward) against spnet_adam_step over the whole buffer: parameters and moments bit for bit
    (the update is element-wise), the l2 penalty to rounding (its partial sums are cut differently); l2 prefix ending inside
    either range, with and without a frozen-element mask."""
    rs = np.random.RandomState(11)
    n = 300_000
    p, g = rs.randn(n).astype(np.float32), (rs.randn(n) * 0.1).astype(np.float32)
    m, v = (rs.randn(n) * 0.01).astype(np.float32), (rs.rand(n) * 0.01).astype(np.float32)
    mask = (rs.rand(n) > 0.3).astype(np.float32)
    lr_dev = torch.full((1,), 2.5e-4, device="cuda")
    for cut, l2n, use_mask in ((200_000, 120_000, False), (200_000, 250_004, True), (4, 2, True), (n - 4, n, False)):
        one = [dev(a) for a in (p, g, m, v)]
        two = [dev(a) for a in (p, g, m, v)]
        md = dev(mask)
        mp = md.data_ptr() if use_mask else None
        sq1, l2a = torch.empty(2048, device="cuda"), torch.zeros(2, device="cuda")
        L.spnet_adam_step(*(t.data_ptr() for t in one), n, l2n, 0.0, 0.9, 0.999, 1e-7, 1e-4, 0.5, mp, sq1.data_ptr(),
                          l2a.data_ptr(), lr_dev.data_ptr(), st())
        parts = [int(L.spnet_adam_parts(cut)), int(L.spnet_adam_parts(n - cut))]
        assert all(0 < q <= 2048 for q in parts)
        sq2, l2b = torch.empty(4096, device="cuda"), torch.zeros(2, device="cuda")
        for lo, hi, so in ((0, cut, 0), (cut, n, parts[0])):
            L.spnet_adam_part(*(t.data_ptr() + 4 * lo for t in two), hi - lo, max(0, min(l2n, hi) - lo), 0.0, 0.9, 0.999, 1e-7,
                              1e-4, 0.5, None if mp is None else mp + 4 * lo, sq2.data_ptr() + 4 * so, lr_dev.data_ptr(), st())
        L.spnet_adam_l2_sum(sq2.data_ptr(), sum(parts), 1e-4, l2b.data_ptr(), st())
        torch.cuda.synchronize()
        for a, b, name in zip(one, two, "pgmv"):
            assert torch.equal(a, b), (name, cut, l2n)
        assert not torch.equal(one[0], dev(p))
        close(l2b[0], 1e-4 * np.sum(p[:l2n].astype(np.float64) ** 2), rtol=1e-5, atol=0)
        close(l2b[0], float(l2a[0]), rtol=2e-6, atol=0)
    assert int(L.spnet_adam_parts(0)) == 0
    with pytest.raises(L.HipError):
        L.spnet_adam_part(one[0].data_ptr(), one[1].data_ptr(), one[2].data_ptr(), one[3].data_ptr(), 6, 0, 0.0, 0.9, 0.999,
                          1e-7, 1e-4, 1.0, None, sq2.data_ptr(), None, st())


def test_adam_many_steps_full_size_with_mask_and_device_lr(L):
    """The optimizer path to rounding, at scale: 3.1 M parameters (ragged vector tail), l2 prefix, frozen-element mask,
    gradient scale (1/world), the step size read from device memory (train_step's path), five consecutive steps against
    the Keras-form oracle -- every element within a few ulp of the fp32 reference after every step."""
    import math
    rs = np.random.RandomState(6)
    n, l2n = 3_100_003 // 4 * 4 + 64, 1_000_000
    p = rs.randn(n).astype(np.float32)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    mask = (rs.rand(n) > 0.25).astype(np.float32)
    pd, md, vd, maskd = dev(p), dev(m), dev(v), dev(mask)
    sq, l2out = torch.empty(2048, device="cuda"), torch.empty(2, device="cuda")
    lr_dev = torch.zeros(1, device="cuda")
    wp, wm, wv = p.copy(), m.copy(), v.copy()
    lr, scale = 2e-4, 0.125
    for t in range(1, 6):
        g = (rs.randn(n) * (10.0 ** rs.uniform(-6, 0, n))).astype(np.float32)      # gradients over six decades
        lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        lr_dev.fill_(float(np.float32(lr_t)))
        want_l2 = 1e-4 * np.sum(wp[:l2n].astype(np.float64) ** 2)
        L.spnet_adam_step(pd.data_ptr(), dev(g).data_ptr(), md.data_ptr(), vd.data_ptr(), n, l2n, 0.0, 0.9, 0.999, 1e-7, 1e-4,
                          scale, maskd.data_ptr(), sq.data_ptr(), l2out.data_ptr(), lr_dev.data_ptr(), st())
        gs = g * np.float32(scale)
        a = R.adam_step(wp[:l2n], gs[:l2n], wm[:l2n], wv[:l2n], t, lr, l2=1e-4)
        b = R.adam_step(wp[l2n:], gs[l2n:], wm[l2n:], wv[l2n:], t, lr)
        np_, nm, nv = (np.concatenate([x, y]) for x, y in zip(a, b))
        keep = mask > 0
        wp, wm, wv = np.where(keep, np_, wp), np.where(keep, nm, wm), np.where(keep, nv, wv)
        close(l2out[0], want_l2, rtol=1e-5, atol=0)
        # a step is at most lr_t per element: agreement to 1e-4 of a step, moments to a few ulp
        assert np.abs(pd.cpu().numpy() - wp).max() <= 1e-4 * lr_t + 4e-7 * np.abs(wp).max()
        # (atol: an element whose 0.9 m + 0.1 g nearly cancels carries the rounding of its terms, ~1e-8 * |g|)
        close(md, wm, rtol=2e-6, atol=1e-8)
        close(vd, wv, rtol=2e-6, atol=1e-15)
    frozen = mask == 0
    assert np.array_equal(pd.cpu().numpy()[frozen], p[frozen])                # frozen elements: bit for bit untouched


def test_dropout_mask_is_reproducible(L):
    x = torch.ones(100000, device="cuda")
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    L.spnet_dropout(x.data_ptr(), y1.data_ptr(), x.numel(), 77, 0.1, None, st())
    sd = torch.tensor([77], dtype=torch.int32, device="cuda")
    L.spnet_dropout(x.data_ptr(), y2.data_ptr(), x.numel(), 1, 0.1, sd.data_ptr(), st())   # seed from device memory
    assert torch.equal(y1, y2)
    keep = (y1 > 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
    vals = torch.unique(y1).cpu().numpy()
    np.testing.assert_allclose(sorted(vals), [0.0, 1.0 / 0.9], rtol=1e-6)


def test_warp_affine_identity_flip_translate(L):
    rs = np.random.RandomState(6)
    N, H, W, C = 2, 12, 16, 3
    x = dev(rs.rand(N, H, W, C))
    out = torch.empty_like(x)
    ident = dev(np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32), (N, 1)))
    L.spnet_warp_affine(x.data_ptr(), out.data_ptr(), N, H, W, C, ident.data_ptr(), st())
    assert torch.equal(out, x)
    # horizontal flip: sx = (W-1) - x ; translate by (+3,-2): dst(x,y) = src(x-3, y+2)
    m = dev(np.array([[-1, 0, W - 1, 0, 1, 0], [1, 0, -3, 0, 1, 2]], np.float32))
    L.spnet_warp_affine(x.data_ptr(), out.data_ptr(), N, H, W, C, m.data_ptr(), st())
    assert torch.equal(out[0].cpu(), x[0].cpu().flip(1))
    want = torch.zeros(H, W, C)
    want[:H - 2, 3:, :] = x[1].cpu()[2:, :W - 3, :]
    assert torch.equal(out[1].cpu(), want)


def test_transpose_batched(L):
    rs = np.random.RandomState(5)
    shapes = [(728, 728), (64, 128), (1536, 2048), (33, 100), (4, 4)]
    srcs = [dev(rs.randn(r, c)) for r, c in shapes]
    dsts = [torch.full((c, r), float("nan"), device="cuda") for r, c in shapes]
    table = torch.tensor([v for s_, d_, (r, c) in zip(srcs, dsts, shapes) for v in (s_.data_ptr(), d_.data_ptr(), r, c)],
                         dtype=torch.int64, device="cuda")
    L.spnet_transpose_batched(table.data_ptr(), len(shapes), max(r for r, _ in shapes), max(c for _, c in shapes), st())
    for s_, d_ in zip(srcs, dsts):
        assert torch.equal(d_, s_.t().contiguous())            # a copy: bit-exact


@pytest.mark.parametrize("M,N,K", [(6144, 728, 728), (1000, 136, 200), (96, 64, 32), (37, 64, 288), (23250, 128, 128)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_gemm_with_batchnorm_backward_blended_into_the_a_operand(L, M, N, K, tile):
    """dX = BN'(g, yp) @ W^T without a BatchNorm-backward pass: spnet_bn_bwd_coeffs* + spnet_gemm_f32_bnblend against
    autograd through BatchNorm (fp64) followed by the product; dy_out must hold exactly the blended operand."""
    rs = np.random.RandomState(M + N + K)
    yp = torch.tensor(rs.randn(M, K) * 1.3 + 0.2, dtype=torch.float64, requires_grad=True)
    gamma = torch.tensor(rs.rand(K) + 0.5, dtype=torch.float64, requires_grad=True)
    beta = torch.tensor(rs.randn(K) * 0.2, dtype=torch.float64, requires_grad=True)
    g = torch.tensor(rs.randn(M, K), dtype=torch.float64)
    Wt = torch.tensor(rs.randn(K, N) / np.sqrt(K), dtype=torch.float64)        # W^T, output-major
    mu, var = yp.mean(0), yp.var(0, unbiased=False)
    y = (yp - mu) * torch.rsqrt(var + 1e-3) * gamma + beta
    y.backward(g)
    dy_ref = yp.grad
    dx_ref = dy_ref @ Wt
    ypd, gd, gad, bed, wtd = (t.detach().float().cuda() for t in (yp, g, gamma, beta, Wt))
    save = torch.cat([mu.detach().float(), torch.rsqrt(var.detach() + 1e-3).float()]).cuda()
    cld = (K + 63) // 64 * 64
    coef = torch.zeros(3 * cld, device="cuda")
    dga, dbe = torch.empty(K, device="cuda"), torch.empty(K, device="cuda")
    ws = torch.empty(L.spnet_bn_ws(M, K), device="cuda")
    L.spnet_bn_bwd_coeffs(ypd.data_ptr(), gd.data_ptr(), M, K, gad.data_ptr(), bed.data_ptr(), save.data_ptr(),
                          save[K:].data_ptr(), dga.data_ptr(), dbe.data_ptr(), coef.data_ptr(), cld, ws.data_ptr(), st())
    close(dga, gamma.grad, rtol=1e-4, atol=1e-4 * np.sqrt(M))
    close(dbe, beta.grad, rtol=1e-4, atol=1e-4 * np.sqrt(M))
    assert float(coef.view(3, cld)[:, K:].abs().max()) == 0.0 if cld > K else True
    dx = torch.full((M, N), float("nan"), device="cuda")
    dyo = torch.full((M, K), float("nan"), device="cuda")
    L.spnet_gemm_f32_bnblend(gd.data_ptr(), ypd.data_ptr(), coef.data_ptr(), cld, K, wtd.data_ptr(), N, dx.data_ptr(), N,
                             M, N, K, tile, dyo.data_ptr(), st())
    close(dyo, dy_ref, rtol=1e-4, atol=3e-5)
    close(dx, dx_ref, rtol=1e-4, atol=3e-5 * np.sqrt(K))
    # the weight gradient's operand is exactly what the data gradient multiplied: recompute the blend on the host
    a, b, c = coef.view(3, cld)[0, :K].cpu(), coef.view(3, cld)[1, :K].cpu(), coef.view(3, cld)[2, :K].cpu()
    host = torch.addcmul(torch.addcmul(c.double(), b.double(), ypd.cpu().double()), a.double(), gd.cpu().double())
    close(dyo, host, rtol=1e-6, atol=1e-6)
    # without dy_out
    dx2 = torch.empty_like(dx)
    L.spnet_gemm_f32_bnblend(gd.data_ptr(), ypd.data_ptr(), coef.data_ptr(), cld, K, wtd.data_ptr(), N, dx2.data_ptr(), N,
                             M, N, K, tile, None, st())
    assert torch.equal(dx, dx2)


def test_gaussian_blur_matches_opencv_semantics(L):
    """cv2.GaussianBlur(img, (k,k), 0): fixed small kernels for k <= 7, reflect-101 borders (scipy 'mirror')."""
    from scipy.ndimage import correlate1d
    rs = np.random.RandomState(2)
    N, H, W = 5, 37, 53
    x = rs.rand(N, H, W).astype(np.float32) * 2 - 1
    ks = np.array([0, 3, 7, 5, 3], np.int32)
    taps = {3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
            7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    want = x.copy().astype(np.float64)
    for n in range(N):
        if ks[n]:
            k = np.array(taps[int(ks[n])])
            want[n] = correlate1d(correlate1d(x[n].astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    xd, kd = dev(x), torch.from_numpy(ks).cuda()
    out = torch.full((N, H, W), float("nan"), device="cuda")
    L.spnet_gaussian_blur(xd.data_ptr(), out.data_ptr(), N, H, W, kd.data_ptr(), st())
    close(out, want, rtol=1e-6, atol=1e-6)
    assert torch.equal(out[0], xd[0])                      # k = 0: the frame passes through unchanged


def test_augmenter_real_blur_flag(L):
    """Default = the reference's no-op blur; real_blur=True blurs exactly the frames whose gate opened, with the same
    RNG consumption (so everything else about the augmented set is unchanged)."""
    import random
    from spnet_amd.augmentation import DeviceAugmenter
    X = torch.tensor(np.random.RandomState(7).rand(24, 64, 80, 1).astype(np.float32) * 2 - 1).cuda()
    outs, rng = [], []
    for flag in (False, True):
        aug = DeviceAugmenter(X, real_blur=flag)
        np.random.seed(11)
        random.seed(11)
        params = aug.draw(list(range(24)))
        o = torch.empty_like(X)
        aug.apply(params, o)
        outs.append(o.cpu().numpy())
        rng.append(np.random.rand())
    assert rng[0] == rng[1]
    ks = params["ksize"]
    assert set(np.unique(ks)) <= {0, 3, 7} and (ks > 0).any() and (ks == 0).any()
    for n in range(24):
        same = np.array_equal(outs[0][n], outs[1][n])
        assert same == (ks[n] == 0), (n, ks[n])


def test_calc_errors_on_device_matches_host(L):
    """diagnostics.calc_errors(device=True): integer counts identical to the host loop (which is pinned by the
    reference's own calc_errors through tests/golden), pixel errors to rounding."""
    from spnet_amd import diagnostics as D
    rs = np.random.RandomState(9)
    N = 37
    Yt = rs.rand(N, 576).astype(np.float32) * 20
    Yp = Yt + rs.randn(N, 576).astype(np.float32)
    Yt[:, 6::8] = (rs.rand(N, 72) > 0.6)
    Yp[:, 6::8] = rs.rand(N, 72)
    Yp[0, 6] = 0.5                    # round-half-even cases of int(round(.))
    Yp[1, 6] = 1.5
    host, devr = D.calc_errors(Yp, Yt), D.calc_errors(Yp, Yt, device=True)
    assert host[:7] == devr[:7] and sum(host[:7]) > 0
    np.testing.assert_allclose(devr[7], host[7], rtol=1e-6)
    assert host[8] == devr[8]


@pytest.mark.parametrize("idx_dtype", [torch.int32, torch.int64])
def test_gather_rows(L, idx_dtype):
    """Batch assembly (the minibatch gather of Model.fit / bench.py): dst[i] = src[index[i]], bit for bit; indices
    outside the set are clamped instead of faulting; rows of any length."""
    rs = np.random.RandomState(4)
    src = dev(rs.randn(37, 6, 10, 2))                     # rows of 120 floats
    idx = torch.tensor(rs.randint(0, 37, 16), dtype=idx_dtype, device="cuda")
    dst = torch.full((16, 6, 10, 2), float("nan"), device="cuda")
    L.gather_rows(src, idx, dst)
    assert torch.equal(dst, src[idx.long()])
    bad = torch.tensor([-3, 99, 5], dtype=idx_dtype, device="cuda")
    d3 = torch.full((3, 6, 10, 2), float("nan"), device="cuda")
    L.gather_rows(src, bad, d3)
    assert torch.equal(d3, src[torch.tensor([0, 36, 5], device="cuda")])
    # rows that are not 16-byte multiples (the reference layout's 331 x 331 x 1 frames) take the 4-byte path
    odd = dev(rs.randn(9, 331, 331, 1))
    i2 = torch.tensor(rs.randint(0, 9, 5), dtype=idx_dtype, device="cuda")
    d2 = torch.full((5, 331, 331, 1), float("nan"), device="cuda")
    L.gather_rows(odd, i2, d2)
    assert torch.equal(d2, odd[i2.long()])
    with pytest.raises(L.HipError):                       # idx_bytes other than 4 | 8
        L.spnet_gather_rows(src.data_ptr(), 37, idx.data_ptr(), 2, dst.data_ptr(), 16, 120, st())
