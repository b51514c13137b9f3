"""-m gpu: every HIP kernel behind the C ABI against a plain PyTorch f32 reference of the same op
(tests/ref_ops.py), on seeded inputs, including ragged / tail / strided / empty-input cases.
Tolerance: outputs are bf16, so |err| <= 2e-2 * max|ref| elementwise and rel-L2 <= 2e-2 (bf16 eps = 7.8e-3)."""
import pytest
import torch

from tests import ref_ops as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def check(got, ref, tol=2e-2):
    got, ref = got.float(), ref.float()
    assert torch.isfinite(got).all()
    denom = ref.abs().max().item() + 1e-12
    assert (got - ref).abs().max().item() <= tol * denom
    if ref.norm() > 0:
        assert ((got - ref).norm() / ref.norm()).item() <= tol


@pytest.fixture(scope="module")
def hip():
    from visualcloze_amd import hip as h
    h.require_gpu()
    return h


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 19, 20, 21, 34, 36])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(128, 128, 64), (200, 192, 128), (37, 64, 256), (513, 264, 384), (1, 512, 256)])
def test_gemm(hip, cfg, epi, shape):
    M, N, K = shape
    a_full = rnd(M, K + 64, seed=1)
    a = a_full[:, :K]                       # strided A (lda != K)
    w, bias = rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3)
    res, gate = rnd(M, N, seed=4), rnd(N, seed=5)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    p = hip.make_problem(a, w, bias, out, res=res if epi == 2 else None, gate=gate if epi == 2 else None)
    hip.gemm(p, epi=epi, tile_cfg=cfg)
    torch.cuda.synchronize()
    check(out, R.gemm_ref(a, w, bias, epi, res, gate))


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 19, 20, 21, 34, 36])
@pytest.mark.parametrize("geom", [(2, 16, 24, 256), (1, 64, 320, 384), (2, 5, 27, 128), (1, 8, 200, 768)])
def test_gemm_qkv_epilogue_writes_v_transposed(hip, cfg, geom):
    """EPI_QKV: q | k columns as EPI_BIAS, the V third transposed into vt[b][h][d][l] (l = joint token index: text rows
    first) and bit-identical to what the plain GEMM would have stored; the grouped two-stream launch of a
    DoubleStreamBlock with batch-strided C rows.  Geometries cover V ranges that start on / off a tile edge of every tile
    shape, row counts that are not multiples of 8 (element-wise path), two batch elements, and the L -> Lpad padding."""
    B, T, N, D = geom
    L, H = T + N, D // 128
    Lp = (L + 63) // 64 * 64
    xi, xt = rnd(B * N, D, seed=1), rnd(B * T, D, seed=2)
    wi, wt = rnd(3 * D, D, scale=D ** -0.5, seed=3), rnd(3 * D, D, scale=D ** -0.5, seed=4)
    bi, bt = rnd(3 * D, seed=5), rnd(3 * D, seed=6)

    def run(epi, vt):
        qkv = torch.full((B * L, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
        kw = dict(c_bstride=L * 3 * D)
        v1 = dict(vt=vt, vt_col0=2 * D, vt_rpb=N, vt_row0=T) if vt is not None else {}
        v2 = dict(vt=vt, vt_col0=2 * D, vt_rpb=T, vt_row0=0) if vt is not None else {}
        hip.gemm([hip.make_problem(xi, wi, bi, qkv[T:], M=B * N, c_rpb=N, **kw, **v1),
                  hip.make_problem(xt, wt, bt, qkv[:T], M=B * T, c_rpb=T, **kw, **v2)], epi=epi, tile_cfg=cfg)
        torch.cuda.synchronize()
        return qkv
    plain = run(hip.EPI_BIAS, None).reshape(B, L, 3 * D)
    assert torch.isfinite(plain.float()).all()
    vt = torch.full((B, H, 128, Lp), 7.0, dtype=torch.bfloat16, device=DEV)
    fused = run(hip.EPI_QKV, vt).reshape(B, L, 3 * D)
    assert torch.equal(fused[..., :2 * D], plain[..., :2 * D])                 # q | k untouched by the mode
    want = plain[..., 2 * D:].reshape(B, L, H, 128).permute(0, 2, 3, 1)        # [B, H, 128, L]
    assert torch.equal(vt[..., :L], want)
    assert bool((vt[..., L:] == 7.0).all())                                    # padding columns are never written
    assert bool(torch.isnan(fused[..., 2 * D:].float()).all())                 # and the V columns of C are not either
    check(plain[0, T:], R.gemm_ref(xi[:N], wi, bi, 0, None, None))             # the plain GEMM itself vs f32 torch
    # vt = NULL: the mode is plain EPI_BIAS
    assert torch.equal(run(hip.EPI_QKV, None).reshape(B, L, 3 * D), plain)


@pytest.mark.parametrize("epi", [0, 2, 4])
@pytest.mark.parametrize("cut", [0, 1, 2, 3, 4])
def test_gemm_split_launch_against_block_round_quantisation(hip, epi, cut):
    """tile_cfg 0 may cut problem 0's rows at a multiple of 256 and run the two parts as two launches on different tile
    shapes (vc_gemm, header).  Forced cuts (k << 8) at every position - including the whole of problem 0 (cut 3 = 700 rows
    rounded up) and beyond it - on a grouped two-stream launch with batch-strided C rows, the in-place gated residual and
    the V^T epilogue: same result as the single launch (GEMM_NO_SPLIT) up to the bias-first / bias-last f32 summation
    order of the two tile families, and the torch reference within bf16 tolerance.  cut 0 = the launcher's own choice."""
    B, T, N, D = 1, 136, 700, 384
    L, H = T + N, D // 128
    NO = 3 * D if epi == 4 else D
    xi, xt = rnd(B * N, D, seed=1), rnd(B * T, D, seed=2)
    wi, wt = rnd(NO, D, scale=D ** -0.5, seed=3), rnd(NO, D, scale=D ** -0.5, seed=4)
    bi, bt = rnd(NO, seed=5), rnd(NO, seed=6)
    gi, gt = rnd(NO, seed=7), rnd(NO, seed=8)
    x0 = rnd(B * L, NO, seed=9)
    Lp = (L + 63) // 64 * 64

    def run(tile_cfg):
        x = x0.clone()
        vt = torch.zeros(B, H, 128, Lp, dtype=torch.bfloat16, device=DEV)
        kw = dict(c_bstride=L * NO)
        ri = dict(res=x[T:], gate=gi) if epi == 2 else {}
        rt = dict(res=x[:T], gate=gt) if epi == 2 else {}
        vi = dict(vt=vt, vt_col0=2 * D, vt_rpb=N, vt_row0=T) if epi == 4 else {}
        vv = dict(vt=vt, vt_col0=2 * D, vt_rpb=T, vt_row0=0) if epi == 4 else {}
        hip.gemm([hip.make_problem(xi, wi, bi, x[T:], M=B * N, c_rpb=N, rows_per_batch=N, **kw, **ri, **vi),
                  hip.make_problem(xt, wt, bt, x[:T], M=B * T, c_rpb=T, rows_per_batch=T, **kw, **rt, **vv)], epi=epi, tile_cfg=tile_cfg)
        torch.cuda.synchronize()
        return x, vt
    one, vt1 = run(hip.GEMM_NO_SPLIT)
    two, vt2 = run(cut << 8)
    cols = 2 * D if epi == 4 else NO
    ref = torch.cat([R.gemm_ref(xt, wt, bt, 2 if epi == 2 else 0, x0[:T], gt), R.gemm_ref(xi, wi, bi, 2 if epi == 2 else 0, x0[T:], gi)])
    check(two[:, :cols], ref[:, :cols])
    check(one[:, :cols], ref[:, :cols])
    assert (two[:, :cols] != one[:, :cols]).float().mean().item() < 2e-3
    if epi == 4:
        check(vt2[0, :, :, :L].permute(2, 0, 1).reshape(L, D), ref[:, 2 * D:])
        assert (vt2 != vt1).float().mean().item() < 2e-3
        assert float(vt2[..., L:].float().abs().sum()) == 0.0


@pytest.mark.parametrize("cfg", [34, 36])
@pytest.mark.parametrize("epi", [0, 1, 3])
@pytest.mark.parametrize("M,N,K", [(4400, 3264, 128), (4400, 3264, 64), (4100, 3100 // 8 * 8, 192), (2100, 6400, 320)])
def test_gemm_persistent_tile_loop_equals_one_block_per_tile(hip, cfg, epi, M, N, K):
    """GEMM_PERSIST + more tiles than CUs on a loader-wave tile: ONE persistent workgroup per CU walks the tiles of its XCD's
    strip and fetches the next tile's W(0), W(1) during the current tile's epilogue (gemm_bf16_kernel PERSIST).  Same bits as one
    workgroup per tile (the default) - for K = 64 (one K-tile: nothing to pipeline), 128, 192, 320 (ring wrap-around),
    tile counts that are not multiples of 8, half-empty edge tiles - and the torch reference within bf16 tolerance."""
    a_full = rnd(M, K + 64, seed=1)
    a = a_full[:, :K]
    w, bias = rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3)
    bm, bn = 256, (128 if cfg == 34 else 192)
    assert ((M + bm - 1) // bm) * ((N + bn - 1) // bn) > hip.device_cus()
    outs = []
    for flags in (0, hip.GEMM_PERSIST, hip.GEMM_PERSIST):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        hip.gemm(hip.make_problem(a, w, bias, out), epi=epi, tile_cfg=cfg | flags)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    check(outs[1], R.gemm_ref(a, w, bias, epi, None, None))


@pytest.mark.parametrize("cfg", [0, 34, 36])
def test_gemm_persistent_grouped_qkv_epilogue(hip, cfg):
    """The persistent loop across a PROBLEM boundary (the next tile belongs to the other stream: other A, W, bias, row
    geometry) with batch-strided C rows and the V^T epilogue, two samples: the DoubleStreamBlock qkv launch."""
    B, T, N, D = 2, 520, 1800, 1280
    L, H = T + N, D // 128
    Lp = (L + 63) // 64 * 64
    xi, xt = rnd(B * N, D, seed=1), rnd(B * T, D, seed=2)
    wi, wt = rnd(3 * D, D, scale=D ** -0.5, seed=3), rnd(3 * D, D, scale=D ** -0.5, seed=4)
    bi, bt = rnd(3 * D, seed=5), rnd(3 * D, seed=6)

    def run(flags):
        qkv = torch.full((B * L, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
        vt = torch.full((B, H, 128, Lp), 7.0, dtype=torch.bfloat16, device=DEV)
        kw = dict(c_bstride=L * 3 * D)
        hip.gemm([hip.make_problem(xi, wi, bi, qkv[T:], M=B * N, c_rpb=N, vt=vt, vt_col0=2 * D, vt_rpb=N, vt_row0=T, **kw),
                  hip.make_problem(xt, wt, bt, qkv[:T], M=B * T, c_rpb=T, vt=vt, vt_col0=2 * D, vt_rpb=T, vt_row0=0, **kw)],
                 epi=hip.EPI_QKV, tile_cfg=cfg | flags)
        torch.cuda.synchronize()
        return qkv, vt
    q0, v0 = run(hip.GEMM_NO_SPLIT)
    q1, v1 = run(hip.GEMM_PERSIST | hip.GEMM_NO_SPLIT)
    assert torch.equal(q0[:, :2 * D], q1[:, :2 * D]) and torch.equal(v0, v1)
    assert torch.isfinite(q1[:, :2 * D].float()).all() and bool(torch.isnan(q1[:, 2 * D:].float()).all())
    want = torch.cat([torch.cat([R.gemm_ref(xt[b * T:(b + 1) * T], wt, bt, 0), R.gemm_ref(xi[b * N:(b + 1) * N], wi, bi, 0)]) for b in range(B)])
    check(q1[:, :2 * D], want[:, :2 * D])
    check(v1[..., :L].permute(0, 3, 1, 2).reshape(B * L, D), want[:, 2 * D:])
    q2, v2 = run(hip.GEMM_PERSIST)                         # the launcher's own plan (may split): bf16-equal
    check(q2[:, :2 * D], want[:, :2 * D])
    # and with head-permuted weights + the key norm in the epilogue (the head tiles leave the epilogue early inside the tile loop)
    perm = hip.qkv_head_permutation(H).to(DEV)
    ks = (1 + 0.1 * rnd(128, seed=8)).to(torch.bfloat16)
    rope = torch.stack([rope_table(L), rope_table(L).flip(0)]).contiguous()
    wip, bip, wtp, btp = wi[perm].contiguous(), bi[perm].contiguous(), wt[perm].contiguous(), bt[perm].contiguous()

    def run_kn(flags):
        qkv = torch.full((B * L, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
        vt = torch.full((B, H, 128, Lp), 7.0, dtype=torch.bfloat16, device=DEV)
        kw = dict(c_bstride=L * 3 * D, vt=vt, vt_col0=2 * D, kn_heads=H, kn_scale=ks, kn_rope=rope)
        hip.gemm([hip.make_problem(xi, wip, bip, qkv[T:], M=B * N, c_rpb=N, vt_rpb=N, vt_row0=T, **kw),
                  hip.make_problem(xt, wtp, btp, qkv[:T], M=B * T, c_rpb=T, vt_rpb=T, vt_row0=0, **kw)], epi=hip.EPI_QKV, tile_cfg=flags)
        torch.cuda.synchronize()
        return qkv, vt
    k0, kv0 = run_kn(0)
    k1, kv1 = run_kn(hip.GEMM_PERSIST)
    assert torch.equal(k0[:, :2 * D], k1[:, :2 * D]) and torch.equal(kv0, kv1) and torch.equal(kv0, v0)
    assert torch.equal(k0[:, :D], q0[:, :D]) and not torch.equal(k0[:, D:2 * D], q0[:, D:2 * D])


@pytest.mark.parametrize("cfg", [0, 1, 2, 4, 5, 19, 34, 36])
@pytest.mark.parametrize("geom", [(2, 16, 24, 2), (1, 64, 320, 3), (2, 5, 27, 1), (1, 136, 700, 6), (1, 520, 1800, 10)])
def test_gemm_head_permuted_qkv_and_norms_in_the_epilogue(hip, cfg, geom):
    """VcGemmProblem.kn_heads: qkv weights whose rows are head-permuted (every 192-column tile = one whole query or key head +
    64 V columns) give the SAME C (q | k | v at their logical columns) and the same V^T as the natural order, on every tile
    shape; with kn_scale / qn_scale the 256x192 epilogue also applies QKNorm + RoPE to the key / query heads - bit-identical to
    GEMM + vc_qknorm_rope_vt(parts = K / Q | K), and with qn_prescale to parts | QPRE (the softmax scale folded into q).
    Two streams with their own scales, batch-strided C rows, two batch elements, row counts off every tile edge."""
    B, T, N, H = geom
    D, L = 128 * H, T + N
    Lp = (L + 63) // 64 * 64
    xi, xt = rnd(B * N, D, seed=1), rnd(B * T, D, seed=2)
    wi, wt = rnd(3 * D, D, scale=D ** -0.5, seed=3), rnd(3 * D, D, scale=D ** -0.5, seed=4)
    bi, bt = rnd(3 * D, seed=5), rnd(3 * D, seed=6)
    ks_i, ks_t = (1 + 0.1 * rnd(128, seed=8)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=9)).to(torch.bfloat16)
    qs_i, qs_t = (1 + 0.1 * rnd(128, seed=10)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=11)).to(torch.bfloat16)
    rope = torch.stack([rope_table(L), rope_table(L).flip(0)][:B]).contiguous()
    perm = hip.qkv_head_permutation(H).to(DEV)
    assert sorted(perm.tolist()) == list(range(3 * D))

    def run(permuted, tile, k=False, q=False, pre=False, with_vt=True):
        qkv = torch.full((B * L, 3 * D), float("nan"), dtype=torch.bfloat16, device=DEV)
        vt = torch.full((B, H, 128, Lp), 7.0, dtype=torch.bfloat16, device=DEV)
        kw = dict(c_bstride=L * 3 * D, **(dict(vt=vt, vt_col0=2 * D) if with_vt else {}))
        fused = k or q
        kn_i = dict(kn_heads=H, kn_scale=ks_i if k else None, qn_scale=qs_i if q else None, qn_prescale=pre, kn_rope=rope if fused else None) if permuted else {}
        kn_t = dict(kn_heads=H, kn_scale=ks_t if k else None, qn_scale=qs_t if q else None, qn_prescale=pre, kn_rope=rope if fused else None) if permuted else {}
        w1, b1, w2, b2 = (wi[perm].contiguous(), bi[perm].contiguous(), wt[perm].contiguous(), bt[perm].contiguous()) if permuted else (wi, bi, wt, bt)
        hip.gemm([hip.make_problem(xi, w1, b1, qkv[T:], M=B * N, c_rpb=N, vt_rpb=N, vt_row0=T, **kw, **kn_i),
                  hip.make_problem(xt, w2, b2, qkv[:T], M=B * T, c_rpb=T, vt_rpb=T, vt_row0=0, **kw, **kn_t)], epi=hip.EPI_QKV, tile_cfg=tile)
        torch.cuda.synchronize()
        return qkv, vt
    tile = cfg | hip.GEMM_NO_SPLIT
    plain, vt0 = run(False, tile)
    permd, vt1 = run(True, tile)
    assert torch.equal(permd[:, :2 * D], plain[:, :2 * D]) and torch.equal(vt1, vt0)
    assert bool(torch.isnan(permd[:, 2 * D:].float()).all()) and torch.isfinite(permd[:, :2 * D].float()).all()
    allc, vtn = run(True, tile, with_vt=False)                     # no vt: V lands in C at its logical columns
    assert torch.equal(allc[:, :2 * D], plain[:, :2 * D]) and bool((vtn == 7.0).all())
    assert torch.equal(allc[:, 2 * D:].reshape(B, L, H, 128).permute(0, 2, 3, 1), vt0[..., :L])

    def prepass(parts):
        want = plain.clone()
        hip.qknorm_rope_vt(want, qs_t, ks_t, rope, vt0.clone(), L, H, q_scale2=qs_i, k_scale2=ks_i, split=T, B=B, parts=parts)
        torch.cuda.synchronize()
        return want
    # (kn_scale / qn_scale force the 256x192 tile whatever cfg asks for)
    for kq, pre, parts in (((True, False), False, hip.QKN_K), ((False, True), False, hip.QKN_Q), ((True, True), False, hip.QKN_Q | hip.QKN_K),
                           ((True, True), True, hip.QKN_Q | hip.QKN_K | hip.QKN_QPRE)):
        fused, vt2 = run(True, tile, k=kq[0], q=kq[1], pre=pre)
        want = prepass(parts)
        assert torch.equal(fused[:, :2 * D], want[:, :2 * D]) and torch.equal(vt2, vt0), (kq, pre)
        assert not torch.equal(fused[:, :2 * D], plain[:, :2 * D])
    # the prescaled queries are the plain ones times 128^-0.5 * log2(e), rounded once
    a, b_ = prepass(hip.QKN_Q | hip.QKN_QPRE)[:, :D].float(), prepass(hip.QKN_Q)[:, :D].float()
    c = 128 ** -0.5 * 1.4426950408889634
    assert (a - b_ * c).abs().max().item() <= 2 ** -7 * (b_.abs().max().item() * c) + 1e-12      # (b_ itself is a rounded value)
    with pytest.raises(hip.VclozeHipError):
        hip.qknorm_rope_vt(plain.clone(), qs_t, ks_t, rope, vt0.clone(), L, H, B=B, parts=hip.QKN_K | hip.QKN_QPRE)


def test_gemm_transpose_detecting(hip):
    """A = I with an asymmetric W: a transposed C write cannot pass."""
    n = 128
    a = torch.eye(n, dtype=torch.bfloat16, device=DEV)
    w = (torch.arange(n * n, device=DEV).reshape(n, n) % 61).to(torch.bfloat16)
    out = hip.linear(a, w)
    torch.cuda.synchronize()
    assert torch.equal(out.float(), w.float().t())


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
def test_gemm_grouped_and_inplace_residual(hip, cfg):
    M1, M2, N, K = 300, 136, 384, 256
    a1, a2 = rnd(M1, K, seed=1), rnd(M2, K, seed=2)
    w1, w2 = rnd(N, K, scale=K ** -0.5, seed=3), rnd(N, K, scale=K ** -0.5, seed=4)
    b1, b2 = rnd(N, seed=5), rnd(N, seed=6)
    x = rnd(M1 + M2, N, seed=7)
    x0 = x.clone()
    g1, g2 = rnd(N, seed=8), rnd(N, seed=9)
    p1 = hip.make_problem(a1, w1, b1, x[M2:], res=x[M2:], gate=g1)
    p2 = hip.make_problem(a2, w2, b2, x[:M2], res=x[:M2], gate=g2)
    hip.gemm([p1, p2], epi=hip.EPI_GATE_RES, tile_cfg=cfg)
    torch.cuda.synchronize()
    ref = torch.cat([R.gemm_ref(a2, w2, b2, 2, x0[:M2], g2), R.gemm_ref(a1, w1, b1, 2, x0[M2:], g1)])
    check(x, ref)


def test_gemm_four_problems_batch_strided_rows(hip):
    """Per-GPU batch of 2: img/txt streams of both samples in one grid; QKV-style output rows land batch-strided in a
    joint [B*L, N] buffer and proj-style A rows are read batch-strided out of it (engine.py double blocks)."""
    B, Nn, T, D, NO = 2, 136, 40, 128, 192
    L = Nn + T
    xi, xt = rnd(B * Nn, D, seed=1), rnd(B * T, D, seed=2)
    wi, wt = rnd(NO, D, scale=D ** -0.5, seed=3), rnd(NO, D, scale=D ** -0.5, seed=4)
    bi, bt = rnd(NO, seed=5), rnd(NO, seed=6)
    joint = torch.full((B * L, NO), float("nan"), dtype=torch.bfloat16, device=DEV)
    ld = joint.stride(0)
    pi = hip.make_problem(xi, wi, bi, joint[T:], M=B * Nn, c_rpb=Nn, c_bstride=L * ld)
    pt = hip.make_problem(xt, wt, bt, joint[:T], M=B * T, c_rpb=T, c_bstride=L * ld)
    hip.gemm([pi, pt], epi=0)
    torch.cuda.synchronize()
    ri, rt = R.gemm_ref(xi, wi, bi, 0), R.gemm_ref(xt, wt, bt, 0)
    ref = torch.cat([torch.cat([rt[b * T:(b + 1) * T], ri[b * Nn:(b + 1) * Nn]]) for b in range(B)])
    check(joint, ref)
    # read the joint rows back batch-strided as A operands, 4 problems (2 samples x 2 streams), gated residual
    w2 = rnd(D, NO, scale=NO ** -0.5, seed=7)
    b2, gates = rnd(D, seed=8), rnd(B, D, seed=9)
    oi, ot = rnd(B * Nn, D, seed=10), rnd(B * T, D, seed=11)
    oi0, ot0 = oi.clone(), ot.clone()
    ps = [hip.make_problem(joint[T:], w2, b2, oi, res=oi, gate=gates, rows_per_batch=Nn, gate_bstride=D, M=B * Nn, a_rpb=Nn, a_bstride=L * ld),
          hip.make_problem(joint[:T], w2, b2, ot, res=ot, gate=gates, rows_per_batch=T, gate_bstride=D, M=B * T, a_rpb=T, a_bstride=L * ld)]
    # the same two problems again as separate per-sample problems -> 4 in one launch must give the same result
    oi2, ot2 = oi0.clone(), ot0.clone()
    ps4 = []
    for b in range(B):
        ps4.append(hip.make_problem(joint[b * L + T:(b + 1) * L], w2, b2, oi2[b * Nn:(b + 1) * Nn], res=oi2[b * Nn:(b + 1) * Nn], gate=gates[b]))
        ps4.append(hip.make_problem(joint[b * L:b * L + T], w2, b2, ot2[b * T:(b + 1) * T], res=ot2[b * T:(b + 1) * T], gate=gates[b]))
    hip.gemm(ps, epi=hip.EPI_GATE_RES)
    hip.gemm(ps4, epi=hip.EPI_GATE_RES)
    torch.cuda.synchronize()
    for b in range(B):
        check(oi[b * Nn:(b + 1) * Nn], R.gemm_ref(ref[b * L + T:(b + 1) * L].to(torch.bfloat16), w2, b2, 2, oi0[b * Nn:(b + 1) * Nn], gates[b]))
        check(ot[b * T:(b + 1) * T], R.gemm_ref(ref[b * L:b * L + T].to(torch.bfloat16), w2, b2, 2, ot0[b * T:(b + 1) * T], gates[b]))
    assert torch.equal(oi, oi2) and torch.equal(ot, ot2)


@pytest.mark.parametrize("S", [2, 3, 8])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(300, 192, 1024), (513, 264, 576), (37, 64, 512), (4096 + 200, 3072, 512), (1, 392, 1536)])
def test_gemm_splitk_remainder(hip, S, epi, M, N, K):
    """VC_GEMM_SPLITK(S): the 256x192 tiles beyond the last whole round of the CUs (all of them below one round) run as S
    K-slices that leave f32 partial tiles in the scratch; the reduce launch sums them in slice order and applies the epilogue.
    Against torch, against the one-pass kernel (same function up to the f32 summation order: <= 1 bf16 ulp on a few elements),
    bit-reproducible, K-slices of unequal length (nk % S != 0), partial m / n tiles, strided A, residual in place."""
    a = rnd(M, K + 64, seed=1)[:, :K]
    w, bias = rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3)
    gate = rnd(N, seed=5)
    x0 = rnd(M, N, seed=4)
    ws = hip.splitk_workspace(DEV)
    outs = []
    for cfg in (hip.GEMM_SPLITK(S), hip.GEMM_SPLITK(S), 36):
        out = x0.clone() if epi == 2 else torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        p = hip.make_problem(a, w, bias, out, res=out if epi == 2 else None, gate=gate if epi == 2 else None)
        hip.gemm(p, epi=epi, tile_cfg=cfg, splitk_ws=ws)
        torch.cuda.synchronize()
        outs.append(out)
    check(outs[0], R.gemm_ref(a, w, bias, epi, x0, gate))
    assert torch.equal(outs[0], outs[1])                      # static assignment, fixed summation order
    d = (outs[0].float() - outs[2].float()).abs()
    assert d.max().item() <= 2 ** -5 * outs[2].float().abs().max().item()     # vs the one-pass kernel: rounding flips only
    if M * N > 10000:
        assert (d > 0).float().mean().item() < 0.1


def test_gemm_splitk_grouped_batch_strided_with_step_counter(hip):
    """Split-K of a GROUPED launch as the DoubleStream blocks issue it: two streams of two samples, batch-strided A rows out of a
    joint buffer, per-sample gates picked by a device step counter, residual in place - and a scratch that is too small is refused."""
    B, Nn, T, D, NO = 2, 520, 136, 768, 192
    L = Nn + T
    joint = rnd(B * L, NO + 64, seed=1)[:, :NO]
    ld = joint.stride(0)
    w2, b2 = rnd(D, NO, scale=NO ** -0.5, seed=7), rnd(D, seed=8)
    gates = rnd(3, B, D, seed=9)                                   # [step][sample][D]
    step = torch.tensor([2], dtype=torch.int32, device=DEV)
    ws = hip.splitk_workspace(DEV)
    res = {}
    for tag, cfg in (("sk", hip.GEMM_SPLITK(3)), ("one", 36)):
        oi, ot = rnd(B * Nn, D, seed=10), rnd(B * T, D, seed=11)
        oi0, ot0 = oi.clone(), ot.clone()
        ps = [hip.make_problem(joint[T:], w2, b2, oi, res=oi, gate=gates[0], rows_per_batch=Nn, gate_bstride=D, M=B * Nn, a_rpb=Nn, a_bstride=L * ld),
              hip.make_problem(joint[:T], w2, b2, ot, res=ot, gate=gates[0], rows_per_batch=T, gate_bstride=D, M=B * T, a_rpb=T, a_bstride=L * ld)]
        hip.gemm(ps, epi=hip.EPI_GATE_RES, tile_cfg=cfg, step_ptr=step, gate_step_stride=B * D, splitk_ws=ws)
        torch.cuda.synchronize()
        res[tag] = (oi, ot)
    for b in range(B):
        check(res["sk"][0][b * Nn:(b + 1) * Nn], R.gemm_ref(joint[b * L + T:(b + 1) * L], w2, b2, 2, oi0[b * Nn:(b + 1) * Nn], gates[2, b]))
        check(res["sk"][1][b * T:(b + 1) * T], R.gemm_ref(joint[b * L:b * L + T], w2, b2, 2, ot0[b * T:(b + 1) * T], gates[2, b]))
    for k in range(2):
        d = (res["sk"][k].float() - res["one"][k].float()).abs().max().item()
        assert d <= 2 ** -5 * res["one"][k].float().abs().max().item()
    small = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    with pytest.raises(hip.VclozeHipError, match="splitk_ws"):
        hip.gemm(hip.make_problem(joint[:T], w2, b2, rnd(T, D)), tile_cfg=hip.GEMM_SPLITK(2), splitk_ws=small)
    with pytest.raises(hip.VclozeHipError, match="splitk_ws"):
        hip.gemm(hip.make_problem(joint[:T], w2, b2, rnd(T, D)), tile_cfg=hip.GEMM_SPLITK(2))
    with pytest.raises(hip.VclozeHipError, match="too short"):
        hip.gemm(hip.make_problem(rnd(64, 128), rnd(64, 128), None, rnd(64, 64)), tile_cfg=hip.GEMM_SPLITK(4), splitk_ws=ws)


@pytest.mark.parametrize("M,K", [(4608, 15360), (1664, 12288)])
def test_gemm_splitk_taken_by_the_cost_model_at_product_shapes(hip, M, K):
    """tile_cfg 0 with a scratch on offer at the shapes where the launcher takes the split (SDEdit stage: 288 tiles = 256 + 32 x 8
    slices; cfg 1: 112 tiles x 2 slices), N = 3072, gate + residual in place: against torch and the one-pass plan."""
    N = 3072
    a, w, bias = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3)
    gate, x0 = rnd(N, seed=5), rnd(M, N, seed=4)
    ws = hip.splitk_workspace(DEV)
    outs = []
    for cfg, wsk in ((0, ws), (0, ws), (hip.GEMM_NO_SPLITK, ws)):
        out = x0.clone()
        hip.gemm(hip.make_problem(a, w, bias, out, res=out, gate=gate), epi=2, tile_cfg=cfg, splitk_ws=wsk)
        torch.cuda.synchronize()
        outs.append(out)
    check(outs[0], R.gemm_ref(a, w, bias, 2, x0, gate))
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], outs[2])              # the split WAS taken (other summation order -> a few 1-ulp flips)
    d = (outs[0].float() - outs[2].float()).abs()
    assert d.max().item() <= 2 ** -5 * outs[2].float().abs().max().item()


@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(777, 1024, 2048), (300, 264, 4096), (2100, 3072, 768), (1, 192, 256), (4352, 3136, 1600)])
def test_gemm_stream_form_of_the_splitk_remainder(hip, epi, shape):
    """VC_GEMM_STREAMK: the remainder tiles' K-iterations dealt out evenly to work items (two segments where a range crosses a
    tile edge, partial tiles summed in K order by the reduce launch) - every epilogue, partial tiles in M and N, items of ~12
    iterations that cross tile edges (3-4 pieces per tile; K = 4096: 6-7, beyond the reducer's four-at-once path), items that are
    whole tiles (K = 768, 256), more tiles than one round (4352 x 3136: 17 x 17 = 289 tiles = 256 whole + 33 streamed), against
    torch, twice bit for bit, and within rounding of the one-pass kernel."""
    M, N, K = shape
    a = rnd(M, K + 64, seed=1)[:, :K]
    w, bias = rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3)
    gate = rnd(N, seed=5)
    x0 = rnd(M, N, seed=4)
    ws = hip.splitk_workspace(DEV)
    outs = []
    for cfg in (hip.GEMM_STREAMK, hip.GEMM_STREAMK, 36):
        out = x0.clone() if epi == 2 else torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        p = hip.make_problem(a, w, bias, out, res=out if epi == 2 else None, gate=gate if epi == 2 else None)
        hip.gemm(p, epi=epi, tile_cfg=cfg, splitk_ws=ws)
        torch.cuda.synchronize()
        outs.append(out)
    check(outs[0], R.gemm_ref(a, w, bias, epi, x0, gate))
    assert torch.equal(outs[0], outs[1])                      # static assignment, fixed summation order
    d = (outs[0].float() - outs[2].float()).abs()
    assert d.max().item() <= 2 ** -5 * outs[2].float().abs().max().item()     # vs the one-pass kernel: rounding flips only
    if M * N > 10000:
        assert (d > 0).float().mean().item() < 0.1


def test_gemm_stream_form_grouped_batch_strided_with_step_counter(hip):
    """The stream form of a GROUPED launch as the DoubleStream blocks issue it (two streams of two samples, batch-strided A rows,
    per-sample gates picked by a device step counter, residual in place): a work item's two segments may belong to DIFFERENT
    problems.  And what it refuses: a scratch that is too small, problems of unequal K."""
    B, Nn, T, D, NO = 2, 520, 136, 768, 192
    L = Nn + T
    joint = rnd(B * L, NO + 64, seed=1)[:, :NO]
    ld = joint.stride(0)
    w2, b2 = rnd(D, NO, scale=NO ** -0.5, seed=7), rnd(D, seed=8)
    gates = rnd(3, B, D, seed=9)
    step = torch.tensor([2], dtype=torch.int32, device=DEV)
    ws = hip.splitk_workspace(DEV)
    res = {}
    for tag, cfg in (("sk", hip.GEMM_STREAMK), ("one", 36)):
        oi, ot = rnd(B * Nn, D, seed=10), rnd(B * T, D, seed=11)
        oi0, ot0 = oi.clone(), ot.clone()
        ps = [hip.make_problem(joint[T:], w2, b2, oi, res=oi, gate=gates[0], rows_per_batch=Nn, gate_bstride=D, M=B * Nn, a_rpb=Nn, a_bstride=L * ld),
              hip.make_problem(joint[:T], w2, b2, ot, res=ot, gate=gates[0], rows_per_batch=T, gate_bstride=D, M=B * T, a_rpb=T, a_bstride=L * ld)]
        hip.gemm(ps, epi=hip.EPI_GATE_RES, tile_cfg=cfg, step_ptr=step, gate_step_stride=B * D, splitk_ws=ws)
        torch.cuda.synchronize()
        res[tag] = (oi, ot)
    for b in range(B):
        check(res["sk"][0][b * Nn:(b + 1) * Nn], R.gemm_ref(joint[b * L + T:(b + 1) * L], w2, b2, 2, oi0[b * Nn:(b + 1) * Nn], gates[2, b]))
        check(res["sk"][1][b * T:(b + 1) * T], R.gemm_ref(joint[b * L:b * L + T], w2, b2, 2, ot0[b * T:(b + 1) * T], gates[2, b]))
    for k in range(2):
        d = (res["sk"][k].float() - res["one"][k].float()).abs().max().item()
        assert d <= 2 ** -5 * res["one"][k].float().abs().max().item()
    small = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    with pytest.raises(hip.VclozeHipError, match="splitk_ws"):
        hip.gemm(hip.make_problem(joint[:T], w2, b2, rnd(T, D)), tile_cfg=hip.GEMM_STREAMK, splitk_ws=small)
    with pytest.raises(hip.VclozeHipError, match="one K"):
        hip.gemm([hip.make_problem(rnd(300, 128), rnd(192, 128), None, rnd(300, 192)), hip.make_problem(rnd(300, 256), rnd(192, 256), None, rnd(300, 192))],
                 tile_cfg=hip.GEMM_STREAMK, splitk_ws=ws)


@pytest.mark.parametrize("M,K", [(6656, 15360), (7424, 12288), (6656, 3072)])
def test_gemm_stream_form_at_product_shapes(hip, M, K):
    """cfg 3 / cfg 5's N = 3072 launches (416 = 256 + 160 tiles, 464 = 256 + 208): the remainder as 256 stream items of 0.625 /
    0.8125 tile each (VC_GEMM_PREFER_STREAMK: the auto plan with the stream form wherever it is eligible; at K = 3072 only with
    VC_GEMM_STREAMK_ANY_K), gate + residual in place: against torch, twice bit for bit, within rounding of the launcher's own plan."""
    N = 3072
    a, w, bias = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3)
    gate, x0 = rnd(N, seed=5), rnd(M, N, seed=4)
    ws = hip.splitk_workspace(DEV)
    flag = hip.GEMM_PREFER_STREAMK | (hip.GEMM_STREAMK_ANY_K if K < 6144 else 0)
    args = hip.GemmArgs()
    outs = []
    for cfg in (flag, flag, hip.GEMM_NO_SPLITK):
        out = x0.clone()
        hip.gemm(hip.make_problem(a, w, bias, out, res=out, gate=gate), epi=2, tile_cfg=cfg, splitk_ws=ws)
        torch.cuda.synchronize()
        outs.append(out)
    check(outs[0], R.gemm_ref(a, w, bias, 2, x0, gate))
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], outs[2])              # the stream form WAS taken (other summation order -> a few 1-ulp flips)
    d = (outs[0].float() - outs[2].float()).abs()
    assert d.max().item() <= 2 ** -5 * outs[2].float().abs().max().item()


@pytest.mark.parametrize("L,H,dh", [(512, 64, 64), (128, 12, 64), (77 + 51, 3, 128), (40, 2, 64)])
def test_gemm_batched_instances_per_head_products(hip, L, H, dh):
    """VcGemmArgs.batch = H: the per-head S_h = Q_h K_h^T and O_h = S_h V_h products of the T5 / CLIP attention (head_dim 64:
    not the fused kernel's) as ONE launch each - instance h reads its dh columns of q / k and writes its [L, L] slab, then
    S_h V_h^T into its dh columns of O - bit-identical to the H separate launches it replaces."""
    q, k, v = rnd(L, H * dh, seed=1), rnd(L, H * dh, seed=2), rnd(L, H * dh, seed=3)
    s1 = torch.full((H * L, L), float("nan"), dtype=torch.bfloat16, device=DEV)
    p = hip.make_problem(q[:, :dh], k[:, :dh], None, s1[:L])
    p.a_zstride, p.w_zstride, p.c_zstride = dh, dh, L * s1.stride(0)
    hip.gemm(p, batch=H)
    s2 = torch.empty_like(s1)
    for h in range(H):
        hip.gemm(hip.make_problem(q[:, h * dh:(h + 1) * dh], k[:, h * dh:(h + 1) * dh], None, s2[h * L:(h + 1) * L]), tile_cfg=1)
    torch.cuda.synchronize()
    assert torch.equal(s1, s2)
    check(s1.view(H, L, L), torch.einsum("lhd,mhd->hlm", q.float().view(L, H, dh), k.float().view(L, H, dh)))
    if L % 64 == 0:
        sm = (torch.softmax(s1.float().view(H, L, L) * dh ** -0.5, -1)).to(torch.bfloat16).view(H * L, L).contiguous()
        vt = v.view(L, H, dh).permute(1, 2, 0).contiguous().view(H * dh, L)
        o1 = torch.full((L, H * dh), float("nan"), dtype=torch.bfloat16, device=DEV)
        p = hip.make_problem(sm[:L], vt[:dh], None, o1[:, :dh])
        p.a_zstride, p.w_zstride, p.c_zstride = L * sm.stride(0), dh * vt.stride(0), dh
        hip.gemm(p, batch=H)
        torch.cuda.synchronize()
        ref = torch.einsum("hlm,mhd->lhd", sm.float().view(H, L, L), v.float().view(L, H, dh)).reshape(L, H * dh)
        check(o1, ref.to(torch.bfloat16).float())
    with pytest.raises(hip.VclozeHipError, match="batch"):
        hip.gemm(p, epi=hip.EPI_GELU, batch=H)


def test_gemm_rejects_bad_arguments(hip):
    a, w = rnd(8, 100), rnd(16, 100)
    with pytest.raises(hip.VclozeHipError):          # K not a multiple of 64
        hip.linear(a, w)
    with pytest.raises(hip.VclozeHipError):          # empty input
        hip.linear(rnd(0, 64), rnd(16, 64))
    with pytest.raises(hip.VclozeHipError):          # dtype
        hip.linear(torch.zeros(8, 64, device=DEV), rnd(16, 64))


@pytest.mark.parametrize("rows,D", [(10, 256), (1000, 3072), (7, 4096), (3, 8)])
def test_ln_modulate(hip, rows, D):
    x = rnd(rows, D, scale=2.0, seed=1) + 0.5
    sh, sc = rnd(D, seed=2), rnd(D, scale=0.3, seed=3)
    out = hip.ln_modulate(x, sh, sc)
    torch.cuda.synchronize()
    check(out, R.ln_modulate_ref(x, sh, sc))


@pytest.mark.parametrize("rows_a,rows_b,D", [(10, 7, 256), (864, 128, 3072), (5, 1, 4096), (3, 0, 64)])
def test_ln_modulate_two_streams_one_launch(hip, rows_a, rows_b, D):
    """img + txt rows of a DoubleStreamBlock in one launch: bit-identical to two single launches; row counts that do
    not fill a 4-row block straddle the stream boundary inside one workgroup."""
    xa, xb = rnd(rows_a, D, scale=2.0, seed=1) + 0.5, rnd(max(rows_b, 1), D, scale=1.5, seed=4) - 0.25
    sha, sca, shb, scb = rnd(2, D, seed=2), rnd(2, D, scale=0.3, seed=3), rnd(D, seed=5), rnd(D, scale=0.3, seed=6)
    oa, ob = torch.full_like(xa, 7.0), torch.full_like(xb, 7.0)
    rpb_a = (rows_a + 1) // 2                  # two "samples" in the first stream: batch b reads modulation row b
    sets = [(xa, sha, sca, oa, rpb_a)] + ([(xb[:rows_b], shb, scb, ob[:rows_b], rows_b)] if rows_b else [])
    hip.ln_modulate2(sets, mod_bstride=D)
    torch.cuda.synchronize()
    ra = hip.ln_modulate(xa, sha, sca, rows_per_batch=rpb_a, mod_bstride=D)
    assert torch.equal(oa, ra)
    check(oa[:rpb_a], R.ln_modulate_ref(xa[:rpb_a], sha[0], sca[0]))
    check(oa[rpb_a:], R.ln_modulate_ref(xa[rpb_a:], sha[1], sca[1]))
    if rows_b:
        assert torch.equal(ob[:rows_b], hip.ln_modulate(xb[:rows_b], shb, scb))
    assert (ob[rows_b:] == 7.0).all()          # nothing written past the second stream


def rope_table(L):
    pos = torch.arange(L, dtype=torch.float64)[:, None] * torch.linspace(0.01, 1.0, 64, dtype=torch.float64)[None]
    return torch.stack([torch.cos(pos), torch.sin(pos)], -1).float().to(DEV).contiguous()


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 7, 8, 12])
@pytest.mark.parametrize("L,H,extra,kv_len,split", [(64, 2, 0, None, 0), (40, 2, 0, None, 16), (200, 3, 256, None, 0),
                                                    (333, 2, 0, 301, 128), (1, 1, 0, None, 0), (1664, 4, 0, None, 512)])
def test_qknorm_rope_vt_and_attention(hip, variant, L, H, extra, kv_len, split):
    ld = 3 * H * 128 + extra
    qkv = rnd(L, ld, seed=7)
    qs, ks = (1 + 0.1 * rnd(128, seed=8)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=9)).to(torch.bfloat16)
    qs2, ks2 = (1 + 0.1 * rnd(128, seed=10)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=11)).to(torch.bfloat16)
    rope = rope_table(L)
    Lpad = (L + 63) // 64 * 64
    vt = torch.full((H, 128, Lpad), float("nan"), dtype=torch.bfloat16, device=DEV)
    qa, ka, vtref = R.qknorm_rope_ref(qkv, qs, ks, rope, H)
    qb, kb, _ = R.qknorm_rope_ref(qkv, qs2, ks2, rope, H)
    qref, kref = torch.cat([qa[:split], qb[split:]]), torch.cat([ka[:split], kb[split:]])
    work = qkv.clone()
    hip.qknorm_rope_vt(work, qs, ks, rope, vt, L, H, q_scale2=qs2, k_scale2=ks2, split=split)
    torch.cuda.synchronize()
    got = work[:, : 3 * H * 128].float().reshape(L, 3, H, 128)
    check(got[:, 0], qref); check(got[:, 1], kref)
    assert torch.equal(vt[:, :, :L].float(), vtref)                     # exact: pure data movement
    assert float(vt[:, :, L:].float().abs().sum()) == 0.0               # padded keys zero-filled
    assert torch.equal(work[:, 2 * H * 128:], qkv[:, 2 * H * 128:])      # v and trailing columns untouched
    out = torch.full((L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    kvl = None if kv_len is None else torch.tensor([kv_len], dtype=torch.int32, device=DEV)
    hip.attention(work, vt, out, L, H, kv_len=kvl, variant=variant)
    torch.cuda.synchronize()
    v = qkv[:, 2 * H * 128: 3 * H * 128].float().reshape(L, H, 128)
    check(out, R.attention_ref(got[:, 0], got[:, 1], v, kv_len))
    if kv_len is not None:
        assert float(out[kv_len:].float().abs().sum()) == 0.0           # pad_input semantics (math.py:96)


@pytest.mark.parametrize("parts", [1, 2, 3])
@pytest.mark.parametrize("L,H,extra,split,B", [(64, 2, 0, 0, 1), (40, 3, 0, 16, 1), (333, 9, 256, 128, 1), (1, 1, 0, 0, 1),
                                               (300, 24, 0, 44, 2), (1664, 10, 0, 512, 1)])
def test_qknorm_rope_rows_only_equals_full_prepass_bitwise(hip, parts, L, H, extra, split, B):
    """Without the V^T part (it comes from the qkv GEMM's epilogue) the q / k rows go through a kernel that reads each
    token's (cos, sin) row once for 8 heads: same arithmetic in the same order as the full pre-pass -> the same bits in the
    selected rows, everything else (unselected q / k, V, trailing columns, vt) untouched."""
    ld = 3 * H * 128 + extra
    qkv = rnd(B * L, ld, seed=7)
    qs, ks = (1 + 0.1 * rnd(128, seed=8)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=9)).to(torch.bfloat16)
    qs2, ks2 = (1 + 0.1 * rnd(128, seed=10)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=11)).to(torch.bfloat16)
    rope = torch.stack([rope_table(L)] + [rope_table(L).flip(0)] * (B - 1)).contiguous() if B > 1 else rope_table(L)
    Lpad = (L + 63) // 64 * 64
    vt = torch.full((B, H, 128, Lpad), 3.0, dtype=torch.bfloat16, device=DEV)
    full, rows = qkv.clone(), qkv.clone()
    hip.qknorm_rope_vt(full, qs, ks, rope, vt.clone(), L, H, q_scale2=qs2, k_scale2=ks2, split=split, B=B)
    hip.qknorm_rope_vt(rows, qs, ks, rope, vt, L, H, q_scale2=qs2, k_scale2=ks2, split=split, B=B, parts=parts)
    torch.cuda.synchronize()
    D = H * 128
    want = qkv.clone()
    if parts & 1:
        want[:, :D] = full[:, :D]
    if parts & 2:
        want[:, D:2 * D] = full[:, D:2 * D]
    assert torch.equal(rows, want)
    assert bool((vt == 3.0).all())


@pytest.mark.parametrize("variant", [8, 12])
@pytest.mark.parametrize("L,H,extra,kv_len,split,B", [(64, 2, 0, None, 0, 1), (40, 2, 0, None, 16, 1), (200, 3, 256, None, 0, 1),
                                                      (333, 2, 0, 301, 128, 1), (1, 1, 0, None, 0, 1), (1664, 4, 0, None, 512, 1),
                                                      (300, 2, 0, None, 44, 2)])
def test_attention_with_in_kernel_query_norm(hip, variant, L, H, extra, kv_len, split, B):
    """Variants 8 / 12 can apply QKNorm + RoPE to the RAW query rows while loading them (q_norm=...): the pre-pass then does
    K and V^T only (parts = QKN_K | QKN_VT) and must leave the q columns untouched.  Same result as the pre-pass route."""
    ld = 3 * H * 128 + extra
    qkv = rnd(B * L, ld, seed=7)
    qs, ks = (1 + 0.1 * rnd(128, seed=8)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=9)).to(torch.bfloat16)
    qs2, ks2 = (1 + 0.1 * rnd(128, seed=10)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=11)).to(torch.bfloat16)
    rope = torch.stack([rope_table(L)] + [rope_table(L).flip(0)] * (B - 1)).contiguous() if B > 1 else rope_table(L)
    Lpad = (L + 63) // 64 * 64
    vt = torch.zeros((B, H, 128, Lpad), dtype=torch.bfloat16, device=DEV)
    kvl = None if kv_len is None else torch.tensor([kv_len] * B, dtype=torch.int32, device=DEV)
    # route 1: everything in the pre-pass
    w1 = qkv.clone()
    hip.qknorm_rope_vt(w1, qs, ks, rope, vt, L, H, q_scale2=qs2, k_scale2=ks2, split=split, B=B)
    o1 = torch.full((B * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.attention(w1, vt, o1, L, H, kv_len=kvl, variant=variant, B=B)
    # route 2: K and V^T in the pre-pass, the queries inside the attention kernel
    w2 = qkv.clone()
    hip.qknorm_rope_vt(w2, qs, ks, rope, vt, L, H, q_scale2=qs2, k_scale2=ks2, split=split, B=B, parts=hip.QKN_K | hip.QKN_VT)
    torch.cuda.synchronize()
    assert torch.equal(w2[:, :H * 128], qkv[:, :H * 128])                  # q columns untouched
    assert torch.equal(w2[:, H * 128:2 * H * 128], w1[:, H * 128:2 * H * 128])
    o2 = torch.full((B * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.attention(w2, vt, o2, L, H, kv_len=kvl, variant=variant, B=B, q_norm=(qs, qs2, split, rope))
    torch.cuda.synchronize()
    check(o2, o1.float(), tol=1e-2)
    # and against the torch reference of the whole chain, per sample
    for b in range(B):
        rb = rope[b] if B > 1 else rope
        x = qkv[b * L:(b + 1) * L]
        qa, ka, _ = R.qknorm_rope_ref(x, qs, ks, rb, H)
        qb_, kb_, _ = R.qknorm_rope_ref(x, qs2, ks2, rb, H)
        qref, kref = torch.cat([qa[:split], qb_[split:]]), torch.cat([ka[:split], kb_[split:]])
        v = x[:, 2 * H * 128: 3 * H * 128].float().reshape(L, H, 128)
        check(o2[b * L:(b + 1) * L], R.attention_ref(qref, kref, v, kv_len))
    with pytest.raises(hip.VclozeHipError):                                # only the one-wave-per-SIMD kernel has it
        hip.attention(w2, vt, o2, L, H, variant=3, B=B, q_norm=(qs, qs2, split, rope))


@pytest.mark.parametrize("variant", [8, 12])
@pytest.mark.parametrize("L,H,extra,kv_len,split,B", [(64, 2, 0, None, 0, 1), (40, 2, 0, None, 16, 1), (200, 3, 256, None, 0, 1),
                                                      (333, 2, 0, 301, 128, 1), (1, 1, 0, None, 0, 1), (1664, 4, 0, None, 512, 1),
                                                      (300, 2, 0, None, 44, 2), (3968, 8, 0, None, 512, 1)])
def test_attention_with_prescaled_queries(hip, variant, L, H, extra, kv_len, split, B):
    """VcAttention.q_prescaled: the q columns already hold QK-normed, rotated queries times 128^-0.5 * log2(e) (the qkv GEMM's
    epilogue with qn_prescale, here the pre-pass with QKN_QPRE - bit-identical, see the GEMM test) and variants 8 / 12 load
    them straight into their MFMA operand registers.  Same function as the in-kernel query norm (one rounding of the scaled,
    rotated value on both routes; they differ in f32 summation order of the RMS only) and as the torch reference."""
    ld = 3 * H * 128 + extra
    qkv = rnd(B * L, ld, seed=7)
    qs, ks = (1 + 0.1 * rnd(128, seed=8)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=9)).to(torch.bfloat16)
    qs2, ks2 = (1 + 0.1 * rnd(128, seed=10)).to(torch.bfloat16), (1 + 0.1 * rnd(128, seed=11)).to(torch.bfloat16)
    rope = torch.stack([rope_table(L)] + [rope_table(L).flip(0)] * (B - 1)).contiguous() if B > 1 else rope_table(L)
    Lpad = (L + 63) // 64 * 64
    vt = torch.zeros((B, H, 128, Lpad), dtype=torch.bfloat16, device=DEV)
    kvl = None if kv_len is None else torch.tensor([kv_len] * B, dtype=torch.int32, device=DEV)
    w1 = qkv.clone()      # route 1: the queries inside the attention kernel
    hip.qknorm_rope_vt(w1, qs, ks, rope, vt, L, H, q_scale2=qs2, k_scale2=ks2, split=split, B=B, parts=hip.QKN_K | hip.QKN_VT)
    o1 = torch.full((B * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.attention(w1, vt, o1, L, H, kv_len=kvl, variant=variant, B=B, q_norm=(qs, qs2, split, rope))
    w2 = qkv.clone()      # route 2: finished, prescaled query rows
    hip.qknorm_rope_vt(w2, qs, ks, rope, vt, L, H, q_scale2=qs2, k_scale2=ks2, split=split, B=B, parts=hip.QKN_Q | hip.QKN_K | hip.QKN_VT | hip.QKN_QPRE)
    o2 = torch.full((B * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.attention(w2, vt, o2, L, H, kv_len=kvl, variant=variant, B=B, q_prescaled=True)
    torch.cuda.synchronize()
    assert torch.equal(w2[:, H * 128:2 * H * 128], w1[:, H * 128:2 * H * 128])
    check(o2, o1.float(), tol=4e-3)
    if L <= 1664:
        for b in range(B):
            rb = rope[b] if B > 1 else rope
            x = qkv[b * L:(b + 1) * L]
            qa, ka, _ = R.qknorm_rope_ref(x, qs, ks, rb, H)
            qb_, kb_, _ = R.qknorm_rope_ref(x, qs2, ks2, rb, H)
            qref, kref = torch.cat([qa[:split], qb_[split:]]), torch.cat([ka[:split], kb_[split:]])
            v = x[:, 2 * H * 128: 3 * H * 128].float().reshape(L, H, 128)
            check(o2[b * L:(b + 1) * L], R.attention_ref(qref, kref, v, kv_len))
    o3 = torch.full((B * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.attention(w2, vt, o3, L, H, kv_len=kvl, variant=variant, B=B, q_prescaled=True, logit_bound=16.65 * 1.5 * 1.5)
    torch.cuda.synchronize()
    check(o3, o2.float(), tol=1e-2)                                        # the bounded-softmax instantiation
    with pytest.raises(hip.VclozeHipError):                                # only the one-wave-per-SIMD kernel has it
        hip.attention(w2, vt, o2, L, H, variant=3, B=B, q_prescaled=True)
    with pytest.raises(hip.VclozeHipError):                                # finished rows are not normalised again
        hip.attention(w2, vt, o2, L, H, variant=variant, B=B, q_prescaled=True, q_norm=(qs, qs2, split, rope))


@pytest.mark.parametrize("L,H,B", [(3968, 24, 1), (4000, 24, 1), (3752, 24, 1), (6656, 24, 1), (2100, 24, 2), (1100, 24, 4),
                                   (2700, 24, 1),      # 264 items: ONE tail item per XCD, cut into 32 pieces of 1-2 tiles (hard boundaries)
                                   (2500, 24, 1)])     # 240 items < 256 workgroups: every item is a tail item, no whole item follows the pieces
def test_attention_stream_form_tail_combined_in_launch(hip, L, H, B):
    """The stream form of the one-wave-per-SIMD kernel (bounded logits + prescaled queries: the product's launches) with the
    tail split at full head count: variant 12 (pieces combined by attn64_merge_kernel) and variant 28 (pieces made FIRST and
    combined at the end of the same launch through flag words of the zero-initialised scratch - VcAttention.variant bit 16)
    must be BIT-IDENTICAL, launch after launch (a launch leaves every flag word zero for the next one), and agree with the
    unsplit kernel (variant 8: another f32 summation order for the tail rows) and with the f32 torch softmax on a sample of
    heads.  Lengths off the 64-key tile and off the 256-query item included."""
    ld = 3 * H * 128
    qkv = rnd(B * L, ld, seed=21)
    qs, ks = torch.ones(128, dtype=torch.bfloat16, device=DEV), torch.ones(128, dtype=torch.bfloat16, device=DEV)
    rope = torch.stack([rope_table(L)] * B).contiguous() if B > 1 else rope_table(L)
    Lpad = (L + 63) // 64 * 64
    vt = torch.zeros((B, H, 128, Lpad), dtype=torch.bfloat16, device=DEV)
    w = qkv.clone()
    hip.qknorm_rope_vt(w, qs, ks, rope, vt, L, H, B=B, parts=hip.QKN_Q | hip.QKN_K | hip.QKN_VT | hip.QKN_QPRE)
    both = {}
    for lb in (16.65, 0.0):            # bounded logits (no running max) and the running-max form of the same stream skeleton
        outs = {}
        for variant in (8, 12, 28, 28, 12, 28):
            o = torch.full((B * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
            hip.attention(w, vt, o, L, H, variant=variant, B=B, q_prescaled=True, logit_bound=lb)
            torch.cuda.synchronize()
            assert torch.isfinite(o.float()).all(), (variant, lb)
            if variant in outs and variant != 8:
                assert torch.equal(o, outs[variant]), f"variant {variant} (logit_bound {lb}) is not reproducible from launch to launch"
            outs[variant] = o
        assert torch.equal(outs[28], outs[12]), lb                              # same pieces, same order of combination
        check(outs[12], outs[8].float(), tol=1e-2)
        both[lb] = outs
    outs = both[16.65]
    check(both[0.0][28], outs[28].float(), tol=1e-2)                            # the same function with and without a running max
    scr = hip.attention_scratch(torch.device(DEV))
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    nflag = (n_cu * 16 + 255) // 256 * 256                                    # the flag words: the END of the scratch, behind every variant's partials
    assert int(scr[-nflag:].to(torch.int32).sum()) == 0                      # every flag word is zero again
    # the f32 function on two heads of the first sample
    x = qkv[:L]
    qn, kn, _ = R.qknorm_rope_ref(x, qs, ks, rope[0] if B > 1 else rope, H)
    v = x[:, 2 * H * 128: 3 * H * 128].float().reshape(L, H, 128)
    for h in (0, H - 1):
        ref = R.attention_ref(qn[:, h:h + 1], kn[:, h:h + 1], v[:, h:h + 1], None)
        check(outs[28][:L, h * 128:(h + 1) * 128], ref)
        check(both[0.0][28][:L, h * 128:(h + 1) * 128], ref)


def test_attention_stream_form_masks_randomised(hip):
    """The PRODUCT's attention launches (stream form: prescaled queries, variants 8 / 12 / 28, bounded logits and the running-max
    template) under per-sample key padding and per-sample masked gaps - gaps that start at key 0, end at kv_len, cover whole
    64-key tiles, sit inside one tile or straddle two, samples whose last tile is partly padding, items cut into tail pieces -
    against the f32 softmax over the live keys (math.py:9-60: masked keys get no weight, masked query rows come back as
    zeros).  Variant 28 must equal variant 12 bit for bit under every mask (same pieces, same order of combination)."""
    g = torch.Generator().manual_seed(4242)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    for it in range(12):
        if it % 3 == 0:                          # enough 256-query items for a tail: H = 24 heads
            L, H, B = ri(1500, 2700), 24, 1 + it % 2
        else:
            L, H, B = ri(65, 1400), ri(1, 4), ri(1, 3)
        ld = 3 * H * 128
        qkv = rnd(B * L, ld, seed=900 + it)
        ones = torch.ones(128, dtype=torch.bfloat16, device=DEV)
        rope1 = rope_table(L)
        rope = torch.stack([rope1] * B).contiguous() if B > 1 else rope1
        Lpad = (L + 63) // 64 * 64
        vt = torch.zeros((B, H, 128, Lpad), dtype=torch.bfloat16, device=DEV)
        w = qkv.clone()
        hip.qknorm_rope_vt(w, ones, ones, rope, vt, L, H, B=B, parts=hip.QKN_Q | hip.QKN_K | hip.QKN_VT | hip.QKN_QPRE)
        kv, gaps = [], []
        live = torch.ones(B, L, dtype=torch.bool, device=DEV)
        for b in range(B):
            k = L if ri(0, 3) == 0 else ri(max(1, L // 2), L)
            mode = ri(0, 4)
            if mode == 0:
                lo, hi = 0, 0                                        # no gap
            elif mode == 1:
                lo, hi = 0, min(k - 1, 64 * ri(1, 3))                 # starts at key 0, whole tiles
            elif mode == 2:
                lo = ri(0, k - 1); hi = k                             # ends at kv_len
            elif mode == 3:
                lo = ri(0, k - 1); hi = min(k, lo + ri(1, 40))        # inside one tile or straddling two
            else:
                lo = ri(0, k - 1); hi = min(k, lo + ri(64, 400))      # several tiles
            if hi - lo >= k:
                lo, hi = 0, 0
            kv.append(k); gaps.append([lo, hi])
            live[b, k:] = False
            live[b, lo:hi] = False
        kvl = torch.tensor(kv, dtype=torch.int32, device=DEV)
        gp = torch.tensor(gaps, dtype=torch.int32, device=DEV)
        heads = sorted({0, H - 1, ri(0, H - 1)})
        refs = []
        for b in range(B):
            qn, kn, _ = R.qknorm_rope_ref(qkv[b * L:(b + 1) * L], ones, ones, rope1, H)
            v = qkv[b * L:(b + 1) * L, 2 * H * 128: 3 * H * 128].float().reshape(L, H, 128)
            sc = torch.einsum("qhd,khd->hqk", qn[:, heads], kn[:, heads]) * 128 ** -0.5
            sc = sc.masked_fill(~live[b][None, None, :], float("-inf"))
            o = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), v[:, heads]) * live[b][:, None, None]
            refs.append(o)
        outs = {}
        for lb in (16.65, 0.0):
            for variant in (8, 12, 28):
                o = torch.full((B * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
                hip.attention(w, vt, o, L, H, variant=variant, B=B, kv_len=kvl, kv_gap=gp, q_prescaled=True, logit_bound=lb)
                torch.cuda.synchronize()
                tag = (it, L, H, B, kv, gaps, variant, lb)
                assert torch.isfinite(o.float()).all(), tag
                ob = o.reshape(B, L, H, 128)
                assert float(ob[~live].float().abs().sum()) == 0.0, tag
                for b in range(B):
                    got = ob[b][:, heads].float()
                    err = ((got - refs[b]).norm() / refs[b].norm()).item()
                    assert err < 1e-2, tag + (b, err)
                outs[(variant, lb)] = o
            assert torch.equal(outs[(28, lb)], outs[(12, lb)]), (it, L, H, B, kv, gaps, lb)
    scr = hip.attention_scratch(torch.device(DEV))
    nflag = (n_cu * 16 + 255) // 256 * 256
    assert int(scr[-nflag:].to(torch.int32).sum()) == 0


@pytest.mark.parametrize("variant", [0, 3, 8, 12])
@pytest.mark.parametrize("L,lo,hi,kv", [(320, 0, 128, 320), (320, 0, 100, 300), (200, 0, 64, 200), (512, 0, 192, 470), (96, 0, 64, 96)])
def test_attention_leading_keys_masked(hip, variant, L, lo, hi, kv):
    """A masked range that STARTS AT KEY 0 and covers whole 64-key tiles: what model.MaskLayout emits for a sample whose
    txt_mask is all zeros (gap = (0, T)).  The online softmax then meets tiles without a single live key before its first
    live one (running max still at its floor); the reference (math.py:9-60) simply attends over the remaining keys and
    zeroes the masked query rows."""
    H = 2
    qkv = rnd(L, 3 * H * 128, seed=31)
    vt = torch.zeros(1, H, 128, (L + 63) // 64 * 64, dtype=torch.bfloat16, device=DEV)
    x = qkv.float().reshape(L, 3, H, 128)
    vt[0, :, :, :L] = x[:, 2].permute(1, 2, 0).to(torch.bfloat16)
    out = torch.full((L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
    kvl = torch.tensor([kv], dtype=torch.int32, device=DEV)
    gap = torch.tensor([[lo, hi]], dtype=torch.int32, device=DEV)
    hip.attention(qkv, vt, out, L, H, kv_len=kvl, variant=variant, kv_gap=gap)
    torch.cuda.synchronize()
    live = torch.ones(L, dtype=torch.bool, device=DEV)
    live[kv:] = False
    live[lo:hi] = False
    s = torch.einsum("qhd,khd->hqk", x[:, 0], x[:, 1]) * 128 ** -0.5
    s[:, :, ~live] = float("-inf")
    o = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), x[:, 2]).reshape(L, H * 128)
    o[~live] = 0
    check(out, R.rb(o))
    assert float(out[~live].float().abs().sum()) == 0.0


@pytest.mark.parametrize("variant", [8, 12])
@pytest.mark.parametrize("L,H,kv,gap", [(64, 2, None, None), (200, 3, None, None), (333, 2, 301, None), (320, 2, 300, (0, 128)),
                                        (512, 2, 470, (100, 230)), (1664, 4, None, None), (3968, 24, None, None), (4000, 3, None, None)])
def test_attention_bounded_logits_needs_no_running_max(hip, variant, L, H, kv, gap):
    """VcAttention.logit_bound: when the caller bounds |q.k| 128^-0.5 log2(e) (QK-normed operands: a property of the norm
    scales) the one-wave-per-SIMD kernel keeps the softmax's reference point at 0 - no row max, no rescale, no (-m) k-step.
    Same function: against the f32 torch softmax and against the running-max path, with key padding, a masked gap (one that
    starts at key 0 too), the tail split, and L off the 64-key tile."""
    g = torch.Generator().manual_seed(41)
    x = torch.randn(L, 3, H, 128, generator=g)
    x[:, :2] = x[:, :2] / x[:, :2].pow(2).mean(-1, keepdim=True).sqrt()          # |q| = |k| = sqrt(128), as after QKNorm
    qkv = x.reshape(L, 3 * H * 128).to(torch.bfloat16).to(DEV)
    x = qkv.float().reshape(L, 3, H, 128)
    vt = torch.zeros(1, H, 128, (L + 63) // 64 * 64, dtype=torch.bfloat16, device=DEV)
    vt[0, :, :, :L] = x[:, 2].permute(1, 2, 0).to(torch.bfloat16)
    s = torch.einsum("qhd,khd->hqk", x[:, 0], x[:, 1]) * 128 ** -0.5
    bound = float(s.abs().max()) * 1.4426950408889634 * 1.01
    assert bound < 17.0
    live = torch.ones(L, dtype=torch.bool, device=DEV)
    kvl = gp = None
    if kv is not None:
        live[kv:] = False
        kvl = torch.tensor([kv], dtype=torch.int32, device=DEV)
    if gap is not None:
        live[gap[0]:gap[1]] = False
        gp = torch.tensor([list(gap)], dtype=torch.int32, device=DEV)
    outs = []
    for lb in (bound, 0.0, 1e4):           # bounded; running max; a bound too large to trust f32 with -> running max again
        o = torch.full((L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
        hip.attention(qkv, vt, o, L, H, kv_len=kvl, variant=variant, kv_gap=gp, logit_bound=lb)
        outs.append(o)
    torch.cuda.synchronize()
    s[:, :, ~live] = float("-inf")
    ref = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), x[:, 2]).reshape(L, H * 128)
    ref[~live] = 0
    check(outs[0], R.rb(ref), tol=1e-2)
    check(outs[1], R.rb(ref), tol=1e-2)
    assert torch.equal(outs[1], outs[2])
    assert float(outs[0][~live].float().abs().sum()) == 0.0
    assert ((outs[0].float() - outs[1].float()).norm() / outs[1].float().norm()).item() < 4e-3


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 7, 8, 12])
def test_attention_softmax_rescale_branch(hip, variant):
    """Force the online-softmax running max to jump late (a spiked key in the LAST tile) and early."""
    L, H = 256, 1
    q = torch.zeros(L, 3 * 128, device=DEV)
    g = torch.Generator().manual_seed(3)
    q.copy_(torch.randn(L, 384, generator=g) * 0.5)
    q[:, 128:256][200] = q[:, 0:128][7] * 40.0          # key 200 aligned with query 7 -> huge logit
    q[:, 128:256][3] = q[:, 0:128][100] * 40.0          # key 3 aligned with query 100
    qkv = q.to(torch.bfloat16)
    vt = qkv[:, 256:].t().contiguous().reshape(1, 128, L)
    out = torch.empty(L, 128, dtype=torch.bfloat16, device=DEV)
    hip.attention(qkv, vt, out, L, H, variant=variant)
    torch.cuda.synchronize()
    x = qkv.float().reshape(L, 3, 1, 128)
    check(out, R.attention_ref(x[:, 0], x[:, 1], x[:, 2]))


def test_elementwise(hip):
    import oracle.flux_oracle as O
    t = torch.tensor([0.0, 0.348, 1.0], device=DEV)
    fr = O.temb_freqs().to(DEV)
    out = torch.empty(3, 256, dtype=torch.bfloat16, device=DEV)
    hip.timestep_embedding(t, fr, out)
    check(out, O.timestep_embedding(t.cpu()).to(DEV), 1e-2)
    g = torch.tensor([30.0], device=DEV)
    out = torch.empty(1, 256, dtype=torch.bfloat16, device=DEV)
    hip.timestep_embedding(g, fr, out, round_t_bf16=True)
    check(out, O.timestep_embedding(g.cpu(), t_is_bf16=True).to(DEV), 1e-2)
    x = rnd(1000, seed=1)
    check(hip.silu(x), R.rb(torch.nn.functional.silu(x.float())))
    a, b, c = rnd(4, 777, seed=2), rnd(777, seed=3), rnd(777, seed=4)
    check(hip.add3(a, b, c), R.rb(R.rb(a.float() + b.float()) + c.float()))      # b, c broadcast over rows
    check(hip.add3(a, b), R.rb(a.float() + b.float()))
    xx, cc = rnd(50, 64, seed=5), rnd(50, 320, seed=6)
    o = torch.empty(50, 384, dtype=torch.bfloat16, device=DEV)
    hip.concat_cols(xx, cc, o)
    assert torch.equal(o, torch.cat([xx, cc], -1))
    xs, v = rnd(999, seed=7), rnd(999, seed=8)
    dts = torch.tensor([0.1, 0.037, 0.2], device=DEV)
    step = torch.tensor([1], dtype=torch.int32, device=DEV)
    ref = R.rb(xs.float() + R.rb(R.rb(torch.tensor(0.037)).item() * (-v.float())))
    hip.euler_step(xs, v, dts, step)
    check(xs, ref, 1e-6)
    hip.step_advance(step)
    assert step.item() == 2
    d = torch.empty_like(a)
    hip.copy(d, a)
    assert torch.equal(d, a)


def test_graph_replay_with_device_step_counter(hip):
    st = torch.cuda.Stream()
    a, w, bias = rnd(256, 128, seed=1), rnd(128, 128, scale=0.1, seed=2), rnd(128, seed=3)
    out = torch.zeros(256, 128, dtype=torch.bfloat16, device=DEV)
    gates, res = rnd(3, 128, seed=4), rnd(256, 128, seed=5)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    p = hip.make_problem(a, w, bias, out, res=res, gate=gates)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        s = st.cuda_stream
        with hip.Graph(s) as g:
            hip.gemm(p, epi=2, step_ptr=step, gate_step_stride=128, stream=s)
            hip.step_advance(step, stream=s)
        for i in range(3):
            g.launch()
            st.synchronize()
            check(out, R.gemm_ref(a, w, bias, 2, res, gates[i]))


def test_attention_randomised_sweep(hip):
    """Seeded sweep over what the parametrised cases do not enumerate: L off every tile size (1 ... 2600), 1-5 heads, batches
    of 1-3 samples with PER-SAMPLE key padding and per-sample masked gaps (gaps that start at key 0, end at kv_len, cover
    whole tiles or sit inside one), row strides wider than 3 H 128, on the 32-queries-per-wave kernel (variant 3), the
    one-wave-per-SIMD kernel (12 = with tail split, 8 = without) and its bounded-logit form - each against the f32 torch
    softmax over the live keys (math.py:9-60: masked keys get no weight, masked query rows come back as zeros)."""
    g = torch.Generator().manual_seed(77)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    for it in range(18):
        L = ri(1, 160) if it % 3 == 0 else ri(161, 2600)
        H, B, extra = ri(1, 5), ri(1, 3), 8 * ri(0, 4) * (it % 2)
        ld = 3 * H * 128 + extra
        x = torch.randn(B * L, 3, H, 128, generator=g)
        x[:, :2] = x[:, :2] / x[:, :2].pow(2).mean(-1, keepdim=True).sqrt()      # |q| = |k| = sqrt(128), as after QKNorm
        qkv = torch.zeros(B * L, ld, dtype=torch.bfloat16, device=DEV)
        qkv[:, :3 * H * 128] = x.reshape(B * L, 3 * H * 128).to(torch.bfloat16).to(DEV)
        xr = qkv[:, :3 * H * 128].float().reshape(B, L, 3, H, 128)
        Lpad = (L + 63) // 64 * 64
        vt = torch.zeros(B, H, 128, Lpad, dtype=torch.bfloat16, device=DEV)
        vt[:, :, :, :L] = xr[:, :, 2].permute(0, 2, 3, 1).to(torch.bfloat16)
        masked = it % 4 != 0
        kvl = gp = None
        live = torch.ones(B, L, dtype=torch.bool, device=DEV)
        if masked:
            kv = [ri(max(1, L // 2), L) for _ in range(B)]
            gaps = []
            for b in range(B):
                lo = 0 if ri(0, 2) == 0 else ri(0, kv[b] - 1)
                hi = kv[b] if ri(0, 5) == 0 and lo > 0 else ri(lo, min(kv[b], lo + ri(0, 300)))
                if hi - lo >= kv[b]:           # never mask every key of a sample
                    hi = lo
                gaps.append([lo, hi])
                live[b, kv[b]:] = False
                live[b, lo:hi] = False
            kvl = torch.tensor(kv, dtype=torch.int32, device=DEV)
            gp = torch.tensor(gaps, dtype=torch.int32, device=DEV) if it % 4 != 1 else None
            if gp is None:
                live[:] = True
                for b in range(B):
                    live[b, kv[b]:] = False
        s = torch.einsum("bqhd,bkhd->bhqk", xr[:, :, 0], xr[:, :, 1]) * 128 ** -0.5
        bound = float(s.abs().max()) * 1.4426950408889634 * 1.01
        s = s.masked_fill(~live[:, None, None, :], float("-inf"))
        ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), xr[:, :, 2]).reshape(B, L, H * 128)
        ref = (ref * live[:, :, None]).reshape(B * L, H * 128)
        for variant, lb in ((3, 0.0), (8, 0.0), (12, 0.0), (12, bound)):
            out = torch.full((B * L, H * 128), float("nan"), dtype=torch.bfloat16, device=DEV)
            hip.attention(qkv, vt, out, L, H, kv_len=kvl, variant=variant, B=B, kv_gap=gp, logit_bound=lb)
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all(), (it, L, H, B, variant, lb)
            err = ((out.float() - ref).norm() / ref.norm()).item()
            assert err < 1e-2, (it, L, H, B, variant, lb, err)
            assert float(out.reshape(B, L, -1)[~live].float().abs().sum()) == 0.0, (it, L, H, B, variant)


def test_gemm_randomised_shape_sweep(hip):
    """Seeded sweep over ragged shapes for the two tiles the cost model picks (128x128, 256x192 + loader waves): M not a
    multiple of any tile, N only a multiple of 8 (bias / staging slices that end mid-tile), K = 64 ... 1024 (1 to 16
    K-tiles: prologue-only, ring wrap-around), with and without bias, every epilogue, plus a 3-problem grouped launch; each
    problem also under a forced split-K plan with a random slice count (slices of unequal length, down to one K-tile)."""
    g = torch.Generator().manual_seed(2024)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    sk_ws = hip.splitk_workspace(DEV)
    for it in range(14):
        M, N, K = ri(1, 700), 8 * ri(1, 60), 64 * ri(1, 16)
        epi = it % 4
        a, w = rnd(M, K, seed=100 + it), rnd(N, K, scale=K ** -0.5, seed=200 + it)
        bias = rnd(N, seed=300 + it) if it % 3 else None
        res, gate = rnd(M, N, seed=400 + it), rnd(N, seed=500 + it)
        S = min(ri(2, 8), K // 64)            # a forced split-K plan on the same problem (skipped when K holds a single K-tile)
        for cfg in (1, 36, 0) + ((hip.GEMM_SPLITK(S),) if S >= 2 else ()):
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            p = hip.make_problem(a, w, bias, out, res=res if epi == 2 else None, gate=gate if epi == 2 else None)
            hip.gemm(p, epi=epi, tile_cfg=cfg, splitk_ws=sk_ws)
            torch.cuda.synchronize()
            check(out, R.gemm_ref(a, w, bias if bias is not None else torch.zeros(N, dtype=torch.bfloat16, device=DEV), epi, res, gate))
    # grouped: three problems of different M / N sharing K
    K = 192
    probs, refs, outs, keep = [], [], [], []     # GemmProblem holds raw pointers: the operands must stay referenced
    for j, (M, N) in enumerate([(300, 200), (17, 456), (260, 8)]):
        a, w, b = rnd(M, K, seed=600 + j), rnd(N, K, scale=K ** -0.5, seed=610 + j), rnd(N, seed=620 + j)
        o = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        keep.append((a, w, b))
        probs.append(hip.make_problem(a, w, b, o)); outs.append(o); refs.append(R.gemm_ref(a, w, b, 0))
    for cfg in (1, 36, hip.GEMM_SPLITK(3)):
        for o in outs:
            o.fill_(float("nan"))
        hip.gemm(probs, epi=0, tile_cfg=cfg, splitk_ws=sk_ws)
        torch.cuda.synchronize()
        for o, r in zip(outs, refs):
            check(o, r)
