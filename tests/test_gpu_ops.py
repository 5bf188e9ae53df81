"""GPU parity of the reference-shaped operator entry points (include/ps_hip.h) against the CPU oracle.

Bar (BASELINE.json north_star): integer/byte results bit-exact (activation quantization); fp32 results within
1e-3 relative — the tests use 2e-5 (tensor-relative), the measured gap of a different fp32 summation order.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def hip():
    from powerserve_amd import hip as h
    return h


@pytest.mark.parametrize("K", [32, 256, 896, 2048, 4096, 4864, 14336])
@pytest.mark.parametrize("vdt", [8, 15])
def test_quantize_act_bit_exact(ctx, oracle, hip, K, vdt):
    if vdt == 15 and K % 256:
        pytest.skip("Q8_K needs K % 256 == 0")
    rng = np.random.default_rng(K + vdt)
    rows = 5
    x = rng.standard_normal((rows, K)).astype(np.float32)
    x[0] *= 0.01
    x[1] *= 30.0
    x[2, : min(K, 256)] = 0.0                      # all-zero block
    x[3, 3] = 7.5; x[3, 9] = -7.5; x[3, 40 % K] = 7.5  # +-max ties: first occurrence decides the sign
    x[4, ::2] = 1e-40                              # denormals
    dx = ctx.to_device(x)
    rs = ctx.L.ps_hip_row_size(vdt, K)
    out = ctx.empty((rows, rs), np.uint8)
    ctx.check(ctx.L.ps_hip_quantize_act(ctx.h, vdt, dx.ptr, K, rows, out.ptr))
    got = out.numpy()
    for r in range(rows):
        want = oracle.from_float(vdt, x[r])
        assert np.array_equal(got[r], want), f"row {r}: {np.flatnonzero(got[r] != want)[:8]}"


@pytest.mark.parametrize("wt", [2, 8, 12, 13, 14])
@pytest.mark.parametrize("K,N", [(256, 64), (512, 96), (896, 130), (2048, 256), (2816, 37), (4096, 128), (4864, 64), (14336, 32)])
@pytest.mark.parametrize("bs", [1, 2, 5])
def test_mul_mat_quant(ctx, oracle, hip, wt, K, N, bs):
    from powerserve_amd import synth
    if wt in (12, 13, 14) and K % 256:
        pytest.skip("K-quants need K % 256 == 0")
    rng = np.random.default_rng(wt * 1000 + K + N + bs)
    w = synth.random_blocks(rng, wt, N, K)
    x = rng.standard_normal((bs, K)).astype(np.float32)
    want = oracle.mul_mat(wt, w, K, N, x)
    W = ctx.upload_weight(wt, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((bs, N))
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
    got = dy.numpy()
    # the GEMV reproduces the reference's AVX2 accumulation order: bit-exact, not just close
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel_err(got, want), np.flatnonzero(got != want)[:8])
    W.free()


@pytest.mark.parametrize("wt", [2, 8, 12, 13, 14])
@pytest.mark.parametrize("K,N", [(4096, 14001), (2048, 9000), (14336, 7001)])
def test_mul_mat_quant_many_row_groups(ctx, oracle, hip, wt, K, N):
    """More row groups than resident workgroups: every workgroup streams several groups (uneven split, partial
    last group), chunks straddle rows, and the chain consumer runs behind the producers for many rounds."""
    from powerserve_amd import synth
    rng = np.random.default_rng(wt * 77 + K + N)
    w = synth.random_blocks(rng, wt, N, K)
    x = rng.standard_normal((1, K)).astype(np.float32)
    want = oracle.mul_mat(wt, w, K, N, x)
    W = ctx.upload_weight(wt, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((1, N))
    for _ in range(3):  # repeated launches: no state may leak between them
        ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
        got = dy.numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel_err(got, want), np.flatnonzero(got != want)[:8])
    W.free()


@pytest.mark.parametrize("wt", [12, 8, 2, 14, 13])
@pytest.mark.parametrize("K,N,bs", [(4096, 520, 128), (2048, 96, 21), (14336, 72, 12), (256, 64, 9), (896, 40, 7)])
def test_mul_mat_batched(ctx, oracle, hip, wt, K, N, bs):
    """Prefill / tree-verify batches: activations quantized once, 8 columns per workgroup, ragged last column group and
    a partial last row group; every column keeps the reference's accumulation order."""
    from powerserve_amd import synth
    if wt in (12, 13, 14) and K % 256:
        pytest.skip("K-quants need K % 256 == 0")
    rng = np.random.default_rng(K + N + bs + wt)
    w = synth.random_blocks(rng, wt, N, K)
    x = rng.standard_normal((bs, K)).astype(np.float32)
    want = oracle.mul_mat(wt, w, K, N, x)
    W = ctx.upload_weight(wt, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((bs, N))
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
    got = dy.numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel_err(got, want), np.argwhere(got != want)[:8])
    W.free()


@pytest.mark.parametrize("K,N,bs", [(4096, 512, 128), (1024, 96, 120), (14336, 64, 113), (2048, 288, 128), (1024, 64, 160), (2048, 96, 200), (1024, 8224, 70), (1024, 32, 12), (4096, 160, 2), (2048, 8224, 16), (1024, 64, 5)])
def test_mul_mat_q4k_chunk_on_matrix_cores(ctx, oracle, hip, K, N, bs):
    """Batches of a Q4_K weight from 2 columns (N % 32 == 0, K % 1024 == 0) take k_gemm4k.hip (17 and more: the wide kernel, fewer: the narrow one): fp16 MFMA contractions of
    exact integers, producer / consumer waves, two accumulator halves per tile -- still bit-for-bit ggml_vec_dot_q4_K_q8_K
    per column, including a ragged last column tile and an item count that is not a multiple of the padding."""
    from powerserve_amd import synth
    rng = np.random.default_rng(K + N + bs)
    w = synth.random_blocks(rng, 12, N, K)
    x = (rng.standard_normal((bs, K)) * rng.uniform(0.1, 30.0, (bs, 1))).astype(np.float32)
    x[min(3, bs - 1), 256:512] = 0.0  # an all-zero super-block (d = 0)
    want = oracle.mul_mat(12, w, K, N, x)
    W = ctx.upload_weight(12, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((bs, N))
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
    got = dy.numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel_err(got, want), np.argwhere(got != want)[:8])
    W.free()


@pytest.mark.parametrize("cfg,kernel", [(8, "gemm4k_par_kernel<8>"), (4, "gemm4k_par_kernel<4>"), (0, None)])
@pytest.mark.parametrize("K,N,bs", [(14336, 64, 12), (4096, 96, 16), (2048, 32, 3), (1024, 64, 5), (3072, 32, 9), (11264, 32, 2)])
def test_mul_mat_q4k_narrow_batch_few_tiles(ctx, oracle, hip, K, N, bs, cfg, kernel):
    """At most 16 columns over few row tiles (the Q / K / V, O and down launches of a tree batch): gemm4k_par_kernel takes the super-blocks of a
    round side by side through the matrix cores and runs the reference's fp32 chains behind them, in order -- with eight or four waves per
    tile, whole and ragged rounds (K / 256 = 56, 16, 8, 4, 12, 44), an odd number of rounds; cfg 0 is round 3's kernel for the same launch."""
    from powerserve_amd import synth
    rng = np.random.default_rng(K + N + bs)
    w = q4k_extreme_blocks(rng, N, K) if K == 4096 else synth.random_blocks(rng, 12, N, K)
    x = (rng.standard_normal((bs, K)) * rng.uniform(0.1, 30.0, (bs, 1))).astype(np.float32)
    x[bs - 1, 256:512] = 0.0
    want = oracle.mul_mat(12, w, K, N, x)
    W = ctx.upload_weight(12, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((bs, N))
    assert ctx.L.ps_hip_debug_set(3, cfg) == 0
    try:
        for _ in range(2):
            ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
            if kernel:
                assert ctx.L.ps_hip_last_matmul_kernel().decode() == kernel
            got = dy.numpy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel_err(got, want), np.argwhere(got != want)[:8])
    finally:
        ctx.L.ps_hip_debug_set(3, 1)
    W.free()


def q4k_extreme_blocks(rng, N, K):
    """Random Q4_K rows with the corner cases the random generator never draws (synth.random_blocks: scales 8..63, mins tied to
    the scales) planted in every row: block_q4_K = d, dmin (fp16), scales[12] (6-bit scale / min pairs), qs[128]."""
    from powerserve_amd import synth
    w = synth.random_blocks(rng, 12, N, K).reshape(N, K // 256, 144)
    f16 = lambda v: np.frombuffer(np.float16(v).tobytes(), dtype=np.uint8)
    nb = K // 256
    for r in range(N):
        b = w[r]
        b[(r + 0) % nb, 4:8] = 0x00; b[(r + 0) % nb, 8:12] = 0xFF; b[(r + 0) % nb, 12:16] = 0xF0   # every scale 0, every min 63
        if nb > 1: b[(r + 1) % nb, 0:2] = f16(0.0)                                               # d = 0
        if nb > 2: b[(r + 2) % nb, 2:4] = f16(0.0)                                               # dmin = 0
        if nb > 3: b[(r + 3) % nb, 4:16] = 0xFF; b[(r + 3) % nb, 16:] = 0xFF                      # scales, mins 63, every nibble 15
        if nb > 4: b[(r + 4) % nb, 4:16] = 0x00                                                  # every scale and min 0
        if nb > 5: b[(r + 5) % nb, 0:2] = f16(-0.0); b[(r + 5) % nb, 2:4] = f16(-1.5)             # d = -0, negative dmin
        if nb > 6: b[(r + 6) % nb, 0:2] = np.array([1, 0], dtype=np.uint8); b[(r + 6) % nb, 16:] = 0x00  # fp16 subnormal d, every nibble 0
        if nb > 7: b[(r + 7) % nb, 0:2] = f16(-3.0); b[(r + 7) % nb, 4:8] = 0x3F                  # negative d, scales 63 / 0 mixed
    return w.reshape(-1)


@pytest.mark.parametrize("K,N,bs", [(4096, 512, 1), (14336, 64, 1), (1024, 2056, 1), (2048, 96, 1), (4096, 160, 2), (1024, 64, 5), (2048, 64, 12), (1024, 8224, 16),
                                    (1024, 96, 40), (4096, 256, 128), (2048, 32, 130)])
def test_mul_mat_q4k_extreme_blocks(ctx, oracle, hip, K, N, bs):
    """Q4_K corner cases through the single-column producer / consumer mat-vec (gemv4, bs = 1), the narrow (2..16 columns) and
    the wide matrix-core mat-muls: scale 0 with min 63, d = 0, dmin = 0, every field at its maximum, all zero, d = -0 and
    negative dmin, an fp16 subnormal d — against quants at +-127, an all-zero super-block and a huge one. Bit for bit."""
    rng = np.random.default_rng(K + N + bs + 44)
    w = q4k_extreme_blocks(rng, N, K)
    x = (rng.standard_normal((bs, K)) * rng.uniform(0.1, 30.0, (bs, 1))).astype(np.float32)
    x[0, 0:512] = np.where(rng.random(512) < 0.5, 1.0, -1.0) * 7.0   # every quant of these super-blocks at +-127
    x[bs - 1, 256:512] = 0.0                                          # an all-zero super-block (Q8_K d = 0)
    x[bs // 2, 512:768] *= 1e4
    want = oracle.mul_mat(12, w, K, N, x)
    assert np.isfinite(want).all()
    W = ctx.upload_weight(12, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((bs, N))
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
    got = dy.numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel_err(got, want), np.argwhere(got != want)[:8])
    W.free()


@pytest.mark.parametrize("K,N,bs", [(4096, 512, 128), (1024, 96, 120), (14336, 64, 113), (2048, 8224, 40), (1024, 64, 160), (1024, 32, 17), (2048, 160, 9)])
def test_mul_mat_q6k_chunk_on_matrix_cores(ctx, oracle, hip, K, N, bs):
    """Q6_K weights (attn_v, ffn_down, output of the Q4_K_M / Q5_K_M mixes), batches from 9 columns: gemm6k_kernel in
    k_gemm4k.hip -- (q6 - 32) x the even / odd parts of the int8 scale as two fp16 MFMA contractions of exact integers --
    bit-for-bit ggml_vec_dot_q6_K_q8_K per column; scales are drawn over the whole int8 range, +-128 x +-32 included."""
    from powerserve_amd import synth
    rng = np.random.default_rng(K + N + bs + 6)
    w = synth.random_blocks(rng, 14, N, K)
    blocks = w.reshape(-1, 210)  # block_q6_K: ql[128], qh[64], scales[16] (int8), d
    blocks[:, 192:208] = rng.integers(-128, 128, (blocks.shape[0], 16), dtype=np.int8).view(np.uint8)
    blocks[0, 192:208] = np.array([-128, 127] * 8, dtype=np.int8).view(np.uint8)  # the extremes against q6 = 0 and 63 everywhere
    blocks[0, 0:128] = 0; blocks[0, 128:192] = 0
    blocks[1, 192:208] = np.array([127, -128] * 8, dtype=np.int8).view(np.uint8)
    blocks[1, 0:192] = 0xff
    x = (rng.standard_normal((bs, K)) * rng.uniform(0.1, 30.0, (bs, 1))).astype(np.float32)
    x[min(3, bs - 1), 256:512] = 0.0
    x[0, 0:512] = np.where(rng.random(512) < 0.5, 1.0, -1.0) * 7.0  # every quant of these super-blocks at +-127
    want = oracle.mul_mat(14, w, K, N, x)
    W = ctx.upload_weight(14, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((bs, N))
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
    got = dy.numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel_err(got, want), np.argwhere(got != want)[:8])
    W.free()


@pytest.mark.parametrize("K,N,bs", [(4096, 512, 128), (1024, 96, 120), (14336, 64, 113), (2048, 8224, 40), (1024, 64, 160), (1024, 32, 12), (2048, 160, 3)])
def test_mul_mat_q5k_batch_on_matrix_cores(ctx, oracle, hip, K, N, bs):
    """Q5_K weights (the Q5_K_M mix), batches from 2 columns: the Q4_K chunk kernels with the Q5_K producer (fifth bit from
    the qh plane; value x scale <= 31 * 63 stays an exact fp16 integer) -- bit-for-bit ggml_vec_dot_q5_K_q8_K per column."""
    from powerserve_amd import synth
    rng = np.random.default_rng(K + N + bs + 5)
    w = synth.random_blocks(rng, 13, N, K)
    blocks = w.reshape(-1, 176)  # block_q5_K: d, dmin, scales[12], qh[32], qs[128]
    blocks[0, 4:16] = 0xff; blocks[0, 16:] = 0xff  # every scale and min 63, every weight 31
    x = (rng.standard_normal((bs, K)) * rng.uniform(0.1, 30.0, (bs, 1))).astype(np.float32)
    x[min(3, bs - 1), 256:512] = 0.0
    x[0, 0:256] = np.where(rng.random(256) < 0.5, 1.0, -1.0) * 7.0  # every quant of this super-block at +-127
    want = oracle.mul_mat(13, w, K, N, x)
    W = ctx.upload_weight(13, w, K, N)
    dx, dy = ctx.to_device(x), ctx.empty((bs, N))
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
    got = dy.numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rel_err(got, want), np.argwhere(got != want)[:8])
    W.free()


@pytest.mark.parametrize("wt", [12, 13, 14])
def test_mul_mat_kquant_random_shapes(ctx, oracle, hip, wt):
    """Seeded sweep over shapes the parametrized cases do not name: column counts 2 .. 200 (narrow kernel, wide kernel with 1 .. 4
    column blocks, ragged last tiles), row counts that leave padded items, K = 1024 .. 4096 in steps of 1024; Q4_K, Q5_K, Q6_K."""
    from powerserve_amd import synth
    rng = np.random.default_rng(1000 + wt)
    for case in range(14):
        K = int(rng.integers(1, 5)) * 1024
        N = int(rng.integers(1, 19)) * 32
        bs = int(rng.choice([2, 3, 7, 9, 15, 16, 17, 31, 33, 48, 63, 64, 65, 100, 127, 129, 161, 200]))
        w = synth.random_blocks(rng, wt, N, K)
        x = (rng.standard_normal((bs, K)) * rng.uniform(0.05, 20.0, (bs, 1))).astype(np.float32)
        want = oracle.mul_mat(wt, w, K, N, x)
        W = ctx.upload_weight(wt, w, K, N)
        dx, dy = ctx.to_device(x), ctx.empty((bs, N))
        ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(dy.tensor()), C.byref(W.tensor()), C.byref(dx.tensor())))
        got = dy.numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (case, K, N, bs, rel_err(got, want), np.argwhere(got != want)[:4])
        W.free()


def test_mul_mat_f32_gqa_views(ctx, oracle, hip):
    """K-cache view x permuted q (norm_attention.cpp:115-129) and V-cache view x kq (:138-147)."""
    rng = np.random.default_rng(7)
    hs, n_kv_heads, r2, n_ctx, n_kv, bs = 64, 2, 3, 48, 33, 4
    n_heads, kvd, dim = n_kv_heads * r2, n_kv_heads * hs, n_kv_heads * r2 * hs
    Kc = rng.standard_normal((n_ctx, kvd)).astype(np.float32)
    q = rng.standard_normal((bs, n_heads, hs)).astype(np.float32)
    dK, dq = ctx.to_device(Kc), ctx.to_device(q)
    kq = ctx.empty((n_heads, bs, n_kv))
    k_view = dK.tensor(ne=[hs, n_kv, n_kv_heads, 1], nb=[4, kvd * 4, hs * 4, hs * 4 * n_kv_heads])
    q_perm = dq.tensor(ne=[hs, bs, n_heads, 1], nb=[4, dim * 4, hs * 4, dim * 4 * bs])
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(kq.tensor()), C.byref(k_view), C.byref(q_perm)))
    want = np.einsum("jgd,bgrd->grbj", Kc[:n_kv].reshape(n_kv, n_kv_heads, hs), q.reshape(bs, n_kv_heads, r2, hs)).reshape(n_heads, bs, n_kv)
    assert rel_err(kq.numpy(), want) < TOL
    want_kq = np.stack([[[oracle.L.pso_vec_dot_f32(hs, Kc[j, (h // r2) * hs:(h // r2 + 1) * hs].ctypes.data, q[i, h].ctypes.data) for j in range(n_kv)] for i in range(bs)] for h in range(n_heads)]).astype(np.float32)
    assert np.array_equal(kq.numpy(), want_kq)  # AVX accumulation order reproduced exactly
    # V (transposed cache [kv_dim][n_ctx]) x p
    Vc = rng.standard_normal((kvd, n_ctx)).astype(np.float32)
    p = rng.random((n_heads, bs, n_kv)).astype(np.float32)
    dV, dp = ctx.to_device(Vc), ctx.to_device(p)
    out = ctx.empty((n_heads, bs, hs))
    v_view = dV.tensor(ne=[n_kv, hs, n_kv_heads, 1], nb=[4, n_ctx * 4, n_ctx * 4 * hs, n_ctx * 4 * hs * n_kv_heads])
    ctx.check(ctx.L.ps_hip_mul_mat(ctx.h, C.byref(out.tensor()), C.byref(v_view), C.byref(dp.tensor())))
    want = np.einsum("gdj,grbj->grbd", Vc[:, :n_kv].reshape(n_kv_heads, hs, n_kv), p.reshape(n_kv_heads, r2, bs, n_kv)).reshape(n_heads, bs, hs)
    assert rel_err(out.numpy(), want) < TOL
    Vn = np.ascontiguousarray(Vc[:, :n_kv])  # (n_kv = 33: one whole block of 32 chains' steps + one leftover position)
    want_pv = np.stack([[[oracle.L.pso_vec_dot_f32(n_kv, Vn[(h // r2) * hs + d].ctypes.data, p[h, i].ctypes.data) for d in range(hs)] for i in range(bs)] for h in range(n_heads)]).astype(np.float32)
    assert np.array_equal(out.numpy(), want_pv)  # the same order here


@pytest.mark.parametrize("dim,eps", [(896, 1e-6), (2048, 1e-5), (4096, 1e-5)])
def test_rms_norm(ctx, oracle, hip, dim, eps):
    rng = np.random.default_rng(dim)
    x = (rng.standard_normal((6, dim)) * 3).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(dim)).astype(np.float32)
    dx, dw, dy = ctx.to_device(x), ctx.to_device(w), ctx.empty((6, dim))
    ctx.check(ctx.L.ps_hip_rms_norm(ctx.h, C.byref(dy.tensor()), C.byref(dx.tensor()), C.byref(dw.tensor()), eps))
    want = oracle.rms_norm(x, w, eps)
    got = dy.numpy()
    assert np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)).max() <= 1  # <= 1 ulp


@pytest.mark.parametrize("fs,af", [(1.0, 1.0), (0.5, 1.0), (0.25, 1.25), (1.0, 0.8)])  # rope_freq_scale, rope_attn_factor (src/core/config.cpp:96,98)
@pytest.mark.parametrize("mode,hs,base", [(0, 64, 1e4), (2, 64, 1e6), (0, 128, 5e5)])
def test_rope_bit_exact(ctx, oracle, hip, mode, hs, base, fs, af):
    from oracle import binding as B
    rng = np.random.default_rng(hs + mode)
    pos = np.array([0, 1, 17, 2047, 4095], dtype=np.int32)
    x = rng.standard_normal((pos.size, 8, hs)).astype(np.float32)
    rp = hip.RopeParams(hs, 4096, base, fs, 0.0, af, 32.0, 0.0, mode)
    dx, dy = ctx.to_device(x), ctx.empty(x.shape)
    ctx.check(ctx.L.ps_hip_rope(ctx.h, C.byref(dy.tensor()), C.byref(dx.tensor()), pos.ctypes.data_as(C.c_void_p), pos.size, C.byref(rp)))
    want = oracle.rope(x, pos, B.RopeParams(hs, 4096, base, fs, 0.0, af, 32.0, 0.0, mode))
    assert np.array_equal(dy.numpy(), want)


@pytest.mark.parametrize("n_kv", [1, 7, 33, 200, 2304])
def test_softmax_ext(ctx, oracle, hip, n_kv):
    rng = np.random.default_rng(n_kv)
    bs, nh = 3, 4
    s = (rng.standard_normal((nh, bs, n_kv)) * 4).astype(np.float32)
    pos = np.array([max(0, n_kv - 3), max(0, n_kv - 2), n_kv - 1], dtype=np.int32)
    mask = np.where(np.arange(n_kv)[None, :] <= pos[:, None], 0.0, -np.inf).astype(np.float32)
    ds, dm, do = ctx.to_device(s), ctx.empty((bs, n_kv)), ctx.empty(s.shape)
    ctx.check(ctx.L.ps_hip_get_mask(ctx.h, C.byref(dm.tensor()), pos.ctypes.data_as(C.c_void_p), bs, None))
    assert np.array_equal(dm.numpy(), mask)
    ctx.check(ctx.L.ps_hip_softmax_ext(ctx.h, C.byref(do.tensor()), C.byref(ds.tensor()), C.byref(dm.tensor()), 0.125, 0.0))
    want = oracle.softmax_ext(s, mask, 0.125)
    got = do.numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # poly exp on groups of 8, libm-exact tail
    assert np.all(got[mask[None].repeat(nh, 0) < 0] == 0.0)


def test_add_dup_silu(ctx, oracle, hip):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((5, 96)).astype(np.float32)
    b = rng.standard_normal((5, 96)).astype(np.float32)
    bias = rng.standard_normal((96,)).astype(np.float32)
    da, db, dbias, do = ctx.to_device(a), ctx.to_device(b), ctx.to_device(bias), ctx.empty(a.shape)
    ctx.check(ctx.L.ps_hip_add(ctx.h, C.byref(do.tensor()), C.byref(da.tensor()), C.byref(db.tensor())))
    assert np.array_equal(do.numpy(), oracle.add(a, b))
    ctx.check(ctx.L.ps_hip_add(ctx.h, C.byref(do.tensor()), C.byref(da.tensor()), C.byref(dbias.tensor())))
    assert np.array_equal(do.numpy(), oracle.add(a, bias))
    g = (rng.standard_normal((3, 500)) * 4).astype(np.float32)
    u = rng.standard_normal((3, 500)).astype(np.float32)
    dg, du, dh = ctx.to_device(g), ctx.to_device(u), ctx.empty(g.shape)
    ctx.check(ctx.L.ps_hip_silu_hadamard(ctx.h, C.byref(dh.tensor()), C.byref(dg.tensor()), C.byref(du.tensor())))
    want = oracle.silu_hadamard(g, u)
    assert np.array_equal(dh.numpy().view(np.uint32), want.view(np.uint32))  # glibc-exact expf
    # dup: permuted [hs, bs, heads] view -> contiguous (PERMUTE+CONT, norm_attention.cpp:149-151)
    hs, bs, nh = 16, 3, 4
    src = rng.standard_normal((nh, bs, hs)).astype(np.float32)  # kqv [hs, bs, n_heads]
    dsrc, ddst = ctx.to_device(src), ctx.empty((bs, nh * hs))
    perm = dsrc.tensor(ne=[hs, nh, bs, 1], nb=[4, hs * bs * 4, hs * 4, hs * bs * nh * 4])
    ctx.check(ctx.L.ps_hip_dup(ctx.h, C.byref(ddst.tensor(ne=[hs * nh, bs, 1, 1])), C.byref(perm)))
    assert np.array_equal(ddst.numpy(), src.transpose(1, 0, 2).reshape(bs, nh * hs))


@pytest.mark.parametrize("wt", [0, 2, 8, 12, 13, 14])
def test_get_embedding(ctx, oracle, hip, wt):
    from powerserve_amd import synth
    rng = np.random.default_rng(wt)
    dim, vocab = 512, 100
    tab = synth.random_blocks(rng, wt, vocab, dim)
    W = ctx.upload_weight(wt, tab, dim, vocab)
    toks = np.array([0, 99, 5, 5, 42], dtype=np.int32)
    out = ctx.empty((toks.size, dim))
    ctx.check(ctx.L.ps_hip_get_embedding(ctx.h, C.byref(out.tensor()), C.byref(W.tensor()), toks.ctypes.data_as(C.c_void_p), toks.size))
    want = oracle.get_embedding(wt, tab, dim, toks)
    assert np.array_equal(out.numpy(), want)
    W.free()


def test_argmax_first_max(ctx, hip):
    x = np.zeros((3, 1000), dtype=np.float32)
    x[0, 17] = 2.0; x[0, 500] = 2.0
    x[1, 999] = 1.0
    dx, out = ctx.to_device(x), ctx.empty((3,), np.int32)
    ctx.check(ctx.L.ps_hip_argmax(ctx.h, dx.ptr, 1000, 3, out.ptr))
    assert out.numpy().tolist() == [17, 999, 0]


def test_errors_are_reported_not_thrown(ctx, hip):
    a = ctx.empty((2, 64))
    t = a.tensor()
    bad = a.tensor(ne=[32, 4, 1, 1])
    rc = ctx.L.ps_hip_mul_mat(ctx.h, C.byref(t), C.byref(t), C.byref(bad))
    assert rc != 0 and b"mul_mat" in ctx.L.ps_hip_last_error(ctx.h)


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("M,N,K,beta", [(300, 512, 256, 0.0), (256, 768, 1024, 1.0), (1, 256, 64, 0.0), (700, 256, 4096, 1.0)])
def test_f16_perf_gemm_variants(ctx, hip, variant, M, N, K, beta):
    """The fp16 perf mode's GEMMs (csrc/perf16.hip; NOT part of the parity path): 128-token tiles through registers (1), 256 tokens x 256 / 128 weight
    rows on the LDS-DMA path (2 / 3) against a k-ordered fp32 reference on synthetic operands in [-1, 1): a ragged last token tile, a single token,
    one k block, the residual form (beta = 1).  The tolerance is the fp32 summation-order difference over K products of magnitude <= 1."""
    assert ctx.L.ps_hip_debug_set(4, variant) == 0
    try:
        us, err = C.c_double(), C.c_double()
        ctx.check(ctx.L.ps_hip_debug_f16_gemm(ctx.h, M, N, K, 1, beta, C.byref(us), C.byref(err)))
        assert err.value <= 2e-6 * K * (1.0 + beta) + 1e-5, (variant, M, N, K, beta, err.value)
    finally:
        ctx.L.ps_hip_debug_set(4, 0)
