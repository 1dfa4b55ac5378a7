"""itermvs_conv2d (hand-written direct convolutions with fused epilogues) vs torch.nn.functional
on the same GPU, for every layer shape of the path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from itermvs_amd import ops as _ops
    return _ops


@pytest.fixture(autouse=True, params=["bf16x3", "fp32"])
def conv_arithmetic(request):
    """every test of this file runs with both forms of the 3x3 layers with more than 8 input channels: the bf16x3 split on the
    bf16 MFMA (conv_tile3.hip, the default) and the exact fp32 MFMA (conv_tile.hip); the other layers have one form"""
    was = ops().MFMA_SPLIT3_DEFAULT
    ops().MFMA_SPLIT3_DEFAULT = request.param == "bf16x3"
    yield request.param
    ops().MFMA_SPLIT3_DEFAULT = was


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-6))


CASES = [
    # cin, cout, k, stride, pad, dil, h, w, n
    (3, 8, 3, 1, 1, 1, 32, 40, 5),      # FeatureNet conv1
    (8, 16, 3, 2, 1, 1, 32, 40, 5),     # layer1.0 stride 2
    (16, 16, 3, 1, 1, 1, 16, 20, 5),
    (32, 48, 3, 2, 1, 1, 16, 24, 2),
    (48, 48, 3, 1, 1, 1, 9, 11, 2),     # odd sizes
    (48, 16, 3, 1, 1, 1, 16, 20, 2),    # output1
    (16, 48, 1, 1, 0, 1, 16, 20, 2),    # inner1 (1x1)
    (8, 8, 3, 1, 1, 1, 16, 20, 10),     # CorrNet conv0
    (16, 32, 3, 2, 1, 1, 8, 10, 10),    # CorrNet conv2
    (8, 1, 3, 1, 1, 1, 16, 20, 10),     # CorrNet conv5 (Cout = 1)
    (43, 32, 3, 1, 2, 2, 16, 20, 1),    # ConvGRU gate, dilation 2
    (32, 64, 1, 1, 0, 1, 16, 20, 1),    # depth head 1x1
    (64, 256, 1, 1, 0, 1, 16, 20, 1),
    (64, 144, 1, 1, 0, 1, 16, 20, 1),   # up-sampling logits
    (16, 1, 1, 1, 0, 1, 8, 10, 64),     # PixelViewWeight 1x1
    # more tiles than resident workgroups: the persistent tile loop, interior fast path, weight reuse
    (8, 16, 3, 1, 1, 1, 256, 320, 3),
    (16, 32, 3, 2, 1, 1, 256, 320, 4),
    (43, 32, 3, 1, 2, 2, 128, 160, 2),  # three channel chunks per tile
]


def packed(wt, fmt):
    return ops().MfmaWeight(wt) if fmt == "mfma" else ops().pack_conv_weight(wt)


@pytest.mark.parametrize("shape", [(16, 48, 1, 32, 40, 3), (32, 48, 1, 18, 22, 2), (16, 32, 3, 64, 96, 2),
                                   (16, 48, 1, 66, 130, 1), (16, 48, 1, 256, 320, 2), (32, 48, 1, 128, 160, 5), (20, 48, 1, 2, 2, 1)])
def test_conv_fused_bilinear_residual(shape):
    """FeatureNet's `F.interpolate(coarse, scale_factor=2, 'bilinear') + inner(x)` (models/net.py:46,49) with the
    up-sampling evaluated in the conv epilogue: identical to bilinear_up + residual add"""
    cin, cout, k, h, w, n = shape
    gen = torch.Generator().manual_seed(cin + cout)
    x = torch.randn((n, cin, h, w), generator=gen).to(DEV)
    wt = (torch.randn((cout, cin, k, k), generator=gen) / (cin * k * k) ** 0.5).to(DEV)
    b = torch.randn((cout,), generator=gen).to(DEV)
    coarse = torch.randn((n, cout, h // 2, w // 2), generator=gen).to(DEV)
    want = F.interpolate(coarse, scale_factor=2, mode="bilinear") + F.conv2d(x, wt, b, padding=k // 2)
    pk = ops().MfmaWeight(wt)
    got = ops().conv2d(x, pk, b, ksize=k, pad=k // 2, add=coarse, add_up2=True)
    assert rel_err(got, want) <= 2e-6
    two_step = ops().conv2d(x, pk, b, ksize=k, pad=k // 2, add=ops().bilinear_up(coarse, 2))
    assert torch.equal(got, two_step)


@pytest.mark.parametrize("shape", [(48, 16, 3, 32, 40, 3), (48, 48, 3, 9, 11, 2), (32, 48, 1, 16, 20, 2), (48, 32, 3, 64, 96, 5)])
def test_conv_channels_last_output(shape):
    """FeatureNet output layers write the channels-last layout the correlation kernels read (+ planar copy)"""
    cin, cout, k, h, w, n = shape
    gen = torch.Generator().manual_seed(cin * 3 + cout)
    x = torch.randn((n, cin, h, w), generator=gen).to(DEV)
    wt = (torch.randn((cout, cin, k, k), generator=gen) / (cin * k * k) ** 0.5).to(DEV)
    b = torch.randn((cout,), generator=gen).to(DEV)
    pk = ops().MfmaWeight(wt)
    planar = ops().conv2d(x, pk, b, ksize=k, pad=k // 2)
    copy = torch.empty_like(planar)
    got = ops().conv2d(x, pk, b, ksize=k, pad=k // 2, channels_last_out=True, out2=copy)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, planar) and torch.equal(copy, planar)
    assert rel_err(got, F.conv2d(x, wt, b, padding=k // 2)) <= 2e-6


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("act", ["none", "relu", "sigmoid", "tanh"])
def test_conv_without_residual_or_bias(case, act):
    """the epilogue variants without the optional operands (separate template instances in the kernels)"""
    cin, cout, k, stride, pad, dil, h, w, n = case
    gen = torch.Generator().manual_seed(cin * 100 + cout + 7)
    x = torch.randn((n, cin, h, w), generator=gen).to(DEV)
    wt = (torch.randn((cout, cin, k, k), generator=gen) / (cin * k * k) ** 0.5).to(DEV)
    want = F.conv2d(x, wt, None, stride=stride, padding=pad, dilation=dil)
    want = {"none": lambda t: t, "relu": F.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](want)
    dense = torch.empty_like(want)
    got = ops().conv2d(x, packed(wt, "mfma"), None, ksize=k, stride=stride, pad=pad, dilation=dil, act=act, out2=dense)
    # sigmoid / tanh compress the value range the error is measured against
    assert rel_err(got, want) <= (2e-6 if act in ("none", "relu") else 2e-5)
    assert torch.equal(dense, got)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("act", ["none", "relu"])
@pytest.mark.parametrize("fmt", ["mfma", "valu"])
def test_conv_matches_torch(case, act, fmt):
    cin, cout, k, stride, pad, dil, h, w, n = case
    gen = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn((n, cin, h, w), generator=gen).to(DEV)
    wt = (torch.randn((cout, cin, k, k), generator=gen) / (cin * k * k) ** 0.5).to(DEV)
    b = torch.randn((cout,), generator=gen).to(DEV)
    want = F.conv2d(x, wt, b, stride=stride, padding=pad, dilation=dil)
    add = torch.randn(want.shape, generator=gen).to(DEV)
    want = want + add
    if act == "relu":
        want = F.relu(want)
    got = ops().conv2d(x, packed(wt, fmt), b, ksize=k, stride=stride, pad=pad, dilation=dil, act=act, add=add)
    assert got.shape == want.shape
    assert rel_err(got, want) <= 2e-6


def test_transposed_conv_with_skip_and_segments():
    """CorrNet conv3/conv4 (itermvs.py:359-363) incl. the per-level weight sets of one launch (matrix-core format; the VALU
    format has no transposed form: the host refuses it)."""
    gen = torch.Generator().manual_seed(5)
    pack = lambda wi: ops().MfmaWeight(wi, transposed=True)
    with pytest.raises(RuntimeError):
        ops().conv2d(torch.zeros((1, 8, 4, 4), device=DEV), ops().pack_conv_weight(torch.zeros((8, 8, 3, 3), device=DEV), transposed=True),
                     None, transposed=True, stride=2, pad=1)
    for cin, cout, h, w in ((32, 16, 8, 10), (16, 8, 7, 9), (32, 16, 32, 40), (16, 8, 64, 80), (8, 8, 5, 33)):
        n = 10
        x = torch.randn((n, cin, h, w), generator=gen).to(DEV)
        ws = [(torch.randn((cin, cout, 3, 3), generator=gen) / (cin * 9) ** 0.5).to(DEV) for _ in range(3)]
        skip = torch.randn((n, cout, 2 * h, 2 * w), generator=gen).to(DEV)
        segs = [(0, 4), (4, 8), (8, 10)]
        want = torch.cat([F.conv_transpose2d(x[a:b], ws[i], stride=2, padding=1, output_padding=1) + skip[a:b]
                          for i, (a, b) in enumerate(segs)])
        got = ops().conv2d(x, [pack(wi) for wi in ws], None, transposed=True,
                           stride=2, pad=1, add=skip, seg_end=[4, 8])
        assert rel_err(got, want) <= 2e-6


def test_split_results_gru_gates():
    """ConvGRU update + reset gates as ONE launch with two results (module.py:61-63): z = sigmoid(conv_z(hx)) and
    r*h = sigmoid(conv_r(hx)) * h, identical to the two separate launches"""
    gen = torch.Generator().manual_seed(11)
    b, hh, ww = 1, 40, 56
    hx = torch.randn((b, 43, hh, ww), generator=gen).to(DEV)
    wz, wr = ((torch.randn((32, 43, 3, 3), generator=gen) / (43 * 9) ** 0.5).to(DEV) for _ in range(2))
    bz, br = (torch.randn((32,), generator=gen).to(DEV) for _ in range(2))
    h = hx[:, :32]
    z1 = ops().conv2d(hx, ops().MfmaWeight(wz), bz, pad=2, dilation=2, act="sigmoid")
    rh1 = ops().conv2d(hx, ops().MfmaWeight(wr), br, pad=2, dilation=2, act="gru_rh", aux1=h)
    rh2 = torch.empty_like(rh1)
    z2 = ops().conv2d(hx, ops().MfmaWeight(torch.cat([wz, wr])), torch.cat([bz, br]), pad=2, dilation=2, act="sigmoid",
                      aux1=h, split=(32, "gru_rh", rh2))
    assert z2.shape == z1.shape and torch.equal(z2, z1) and torch.equal(rh2, rh1)
    want_z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=2, dilation=2))
    want_rh = torch.sigmoid(F.conv2d(hx, wr, br, padding=2, dilation=2)) * h
    assert rel_err(z2, want_z) <= 2e-5 and rel_err(rh2, want_rh) <= 2e-5


@pytest.mark.parametrize("fmt", ["mfma", "valu"])
def test_segmented_conv_and_channel_slices(fmt):
    gen = torch.Generator().manual_seed(6)
    n, h, w = 10, 12, 16
    x = torch.randn((n, 8, h, w), generator=gen).to(DEV)
    ws = [(torch.randn((1, 8, 3, 3), generator=gen) / 8).to(DEV) for _ in range(3)]
    bs = [torch.randn((1,), generator=gen).to(DEV) for _ in range(3)]
    want = torch.cat([F.conv2d(x[a:b], ws[i], bs[i], padding=1) for i, (a, b) in enumerate(((0, 4), (4, 8), (8, 10)))])
    # write the ten score planes straight into channels 33..42 of two [1,43,H,W] buffers
    hx = torch.zeros((1, 43, h, w), device=DEV)
    hx2 = torch.zeros((1, 43, h, w), device=DEV)
    out = hx[0, 33:43].unsqueeze(1)
    ops().conv2d(x, [packed(wi, fmt) for wi in ws], bs, seg_end=[4, 8], out=out, out2=hx2[0, 33:43].unsqueeze(1))
    assert rel_err(hx[0, 33:43], want[:, 0]) <= 2e-6 and rel_err(hx2[0, 33:43], want[:, 0]) <= 2e-6
    assert float(hx[:, :33].abs().max()) == 0.0
    # input given as a channel slice with a wider batch stride
    wide = torch.randn((2, 43, h, w), generator=gen).to(DEV)
    wt = (torch.randn((16, 32, 3, 3), generator=gen) / 17).to(DEV)
    want = F.conv2d(wide[:, :32], wt, padding=2, dilation=2)
    got = ops().conv2d(wide[:, :32], packed(wt, fmt), None, pad=2, dilation=2)
    assert rel_err(got, want) <= 2e-6


@pytest.mark.parametrize("fmt", ["mfma", "valu"])
def test_gru_fused_epilogues(fmt):
    """module.py:59-66 with the gate math inside the conv epilogues."""
    gen = torch.Generator().manual_seed(7)
    b, hh, ww = 2, 16, 20
    h = torch.tanh(torch.randn((b, 32, hh, ww), generator=gen)).to(DEV)
    x = torch.randn((b, 11, hh, ww), generator=gen).to(DEV)
    wz, wr, wq = [(torch.randn((32, 43, 3, 3), generator=gen) / 20).to(DEV) for _ in range(3)]
    bz, br, bq = [torch.randn((32,), generator=gen).to(DEV) * 0.1 for _ in range(3)]
    hx = torch.cat([h, x], 1).contiguous()
    z = torch.sigmoid(F.conv2d(hx, wz, bz, padding=2, dilation=2))
    r = torch.sigmoid(F.conv2d(hx, wr, br, padding=2, dilation=2))
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), wq, bq, padding=2, dilation=2))
    want = (1 - z) * h + z * q
    o = ops()
    hx2 = hx.clone()
    zbuf = o.conv2d(hx, packed(wz, fmt), bz, pad=2, dilation=2, act="sigmoid")
    o.conv2d(hx, packed(wr, fmt), br, pad=2, dilation=2, act="gru_rh", aux1=hx[:, :32], out=hx2[:, :32])
    assert rel_err(hx2[:, :32], r * h) <= 5e-6
    hidden = torch.empty((b, 32, hh, ww), device=DEV)
    o.conv2d(hx2, packed(wq, fmt), bq, pad=2, dilation=2, act="gru_out", aux1=hx[:, :32], aux2=zbuf,
             out=hx[:, :32], out2=hidden)
    assert rel_err(hidden, want) <= 5e-6 and rel_err(hx[:, :32], want) <= 5e-6
    assert torch.equal(hx[:, 32:], x)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_channels_last_output_in_16bit_storage(dtype):
    """FeatureNet's output convolutions can write the pyramid in fp16 / bf16 (out_layout 2 / 3): the fp32 result rounded to
    nearest even, bit for bit what torch's cast gives on the fp32 channels-last output; NaN / inf survive"""
    gen = torch.Generator().manual_seed(4)
    x = torch.randn((3, 48, 20, 28), generator=gen).to(DEV)
    x[0, 0, 0, 0] = 3.0e38
    wt = torch.randn((32, 48, 3, 3), generator=gen).to(DEV) * 0.1
    bias = torch.randn((32,), generator=gen).to(DEV)
    pk = ops().MfmaWeight(wt)
    o32 = torch.empty((3, 32, 20, 28), device=DEV, memory_format=torch.channels_last)
    o16 = torch.empty((3, 32, 20, 28), device=DEV, dtype=dtype, memory_format=torch.channels_last)
    planar = torch.empty((3, 32, 20, 28), device=DEV)
    ops().conv2d(x, pk, bias, channels_last_out=True, out=o32)
    ops().conv2d(x, pk, bias, channels_last_out=True, out=o16, out2=planar)
    assert o16.dtype == dtype and torch.equal(o16, o32.to(dtype))
    assert torch.equal(planar, o32.contiguous())                 # the planar copy stays fp32
    with pytest.raises(RuntimeError):
        ops().conv2d(x, pk, bias, out=torch.empty((3, 32, 20, 28), device=DEV, dtype=dtype))


@pytest.mark.parametrize("case", [(10, 128, 160, (1, 2, 3), (4, 8)), (32, 64, 80, (3,), ()), (5, 36, 52, (2, 1), (3,)), (2, 8, 4, (1,), ())])
def test_corrnet_one_launch_matches_torch(case):
    """itermvs_corrnet (the whole U-Net of itermvs.py:352-381 per 32x32 tile in LDS, halos recomputed, zero padding of
    every intermediate reproduced at the image border) against the torch layer chain, with per-segment weight sets, ragged
    sizes (not multiples of the tile) and maps smaller than a tile; outputs may land in channel slices of wider buffers"""
    from conftest import load_weights
    m, h, w, levels, seg_end = case
    wts = {k: v.to(DEV) for k, v in load_weights("dtu").items() if "corr_conv1" in k}
    gen = torch.Generator().manual_seed(m * h + w)
    x = torch.randn((m, 8, h, w), generator=gen).to(DEV)
    bounds = [0] + list(seg_end) + [m]
    want = []
    for i, l in enumerate(levels):
        p = f"iter_mvs.evaluation.corr_conv1.{l - 1}."
        xi = x[bounds[i]:bounds[i + 1]]
        c0 = F.relu(F.conv2d(xi, wts[p + "conv0.conv.weight"], padding=1))
        c1 = F.relu(F.conv2d(c0, wts[p + "conv1.conv.weight"], stride=2, padding=1))
        c2 = F.relu(F.conv2d(c1, wts[p + "conv2.conv.weight"], stride=2, padding=1))
        u1 = c1 + F.conv_transpose2d(c2, wts[p + "conv3.weight"], stride=2, padding=1, output_padding=1)
        u0 = c0 + F.conv_transpose2d(u1, wts[p + "conv4.weight"], stride=2, padding=1, output_padding=1)
        want.append(F.conv2d(u0, wts[p + "conv5.weight"], wts[p + "conv5.bias"], padding=1))
    want = torch.cat(want)
    # (the file's fixture runs this test twice: conv0 on the bf16 matrix instruction with the exact three-term split, and in fp32)
    split3 = ops().MFMA_SPLIT3_DEFAULT
    packs = [ops().pack_corrnet_weights(wts, f"iter_mvs.evaluation.corr_conv1.{l - 1}.", split3=split3) for l in levels]
    got = ops().corrnet(x, packs, seg_end)
    assert got.shape == want.shape and rel_err(got, want) <= 3e-6, rel_err(got, want)
    if split3:                                # any alignment of the input: same bits
        buf = torch.zeros((x.numel() + 4,), device=DEV)
        xo = buf[1:1 + x.numel()].view(x.shape)
        xo.copy_(x)
        assert torch.equal(ops().corrnet(xo, packs, seg_end), got)
        with pytest.raises(RuntimeError):     # one arithmetic per launch
            ops().corrnet(x, [packs[0], ops().pack_corrnet_weights(wts, "iter_mvs.evaluation.corr_conv1.0.", split3=False)][:max(2, len(packs))], (1,))
    if m == 10:                               # the GRU input buffers take the ten score planes in place
        wide = torch.zeros((1, 43, h, w), device=DEV)
        wide2 = torch.zeros((1, 43, h, w), device=DEV)
        ops().corrnet(x, packs, seg_end, out=wide[0, 33:].unsqueeze(1), out2=wide2[0, 33:].unsqueeze(1))
        assert torch.equal(wide[0, 33:], got[:, 0]) and torch.equal(wide2, wide) and float(wide[:, :33].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops().corrnet(torch.zeros((1, 8, 6, 8), device=DEV), packs[:1])


def test_relu_dot_epilogue_is_conv_relu_conv1x1():
    """act='relu_dot' (PixelViewWeight, itermvs.py:337-346): conv3x3 8 -> 16, ReLU and the 1x1 layer to one channel in
    one launch, the 16-channel tensor never stored"""
    gen = torch.Generator().manual_seed(9)
    x = torch.randn((24, 8, 23, 37), generator=gen).to(DEV)
    w0 = (torch.randn((16, 8, 3, 3), generator=gen) * 0.2).to(DEV)
    w1 = torch.randn((1, 16, 1, 1), generator=gen).to(DEV)
    b1 = torch.randn((1,), generator=gen).to(DEV)
    want = F.conv2d(F.relu(F.conv2d(x, w0, padding=1)), w1, b1)
    got = ops().conv2d(x, ops().MfmaWeight(w0), None, act="relu_dot", aux1=torch.cat([w1.reshape(-1), b1]))
    assert got.shape == want.shape == (24, 1, 23, 37) and rel_err(got, want) <= 2e-6
    with pytest.raises(RuntimeError):
        ops().conv2d(x, ops().MfmaWeight(w0), None, act="relu_dot", aux1=w1.reshape(-1))
    # 32 channels, dilation 2, sigmoid: the confidence head (itermvs.py:147-151,198) in one launch
    h = torch.randn((2, 32, 37, 50), generator=gen).to(DEV)
    c0 = (torch.randn((32, 32, 3, 3), generator=gen) * 0.1).to(DEV)
    c1 = torch.randn((1, 32, 1, 1), generator=gen).to(DEV)
    want = torch.sigmoid(F.conv2d(F.relu(F.conv2d(h, c0, padding=2, dilation=2)), c1, b1))
    got = ops().conv2d(h, ops().MfmaWeight(c0), None, pad=2, dilation=2, act="relu_dot_sigmoid", aux1=torch.cat([c1.reshape(-1), b1]))
    assert got.shape == want.shape == (2, 1, 37, 50) and float((got - want).abs().max()) <= 2e-6


@pytest.mark.parametrize("case", [(5, 512, 640), (2, 96, 160), (1, 50, 70), (3, 16, 64), (1, 7, 5)])
def test_stem_one_launch_matches_torch(case):
    """itermvs_stem (FeatureNet.conv1 + layer1[0].conv1 / .downsample, net.py:13-14,39-40, fea0 kept in LDS) against the
    torch layer chain with BatchNorm folded: full cfg-1 size, ragged sizes (odd, not multiples of the 8 x 32 tile) and an
    image smaller than a tile; zero padding of fea0 at the image border must be reproduced"""
    from conftest import load_weights
    from itermvs_amd.engine import fold_batchnorm
    m, h, w = case
    wts = {k: v.to(DEV) for k, v in load_weights("dtu").items() if k.startswith("feature_net.")}
    (w0, b0), (w1, b1), (wd, bd) = [fold_batchnorm(wts, "feature_net." + n) for n in ("conv1.", "layer1.0.conv1.", "layer1.0.downsample.")]
    gen = torch.Generator().manual_seed(h * w + m)
    x = torch.randn((m, 3, h, w), generator=gen).to(DEV)
    f0 = F.relu(F.conv2d(x, w0, b0, padding=1))
    want_y = F.relu(F.conv2d(f0, w1, b1, stride=2, padding=1))
    want_sc = F.conv2d(f0, wd, bd, stride=2, padding=1)
    y, sc = ops().stem(x, *ops().pack_stem_weights(w0, b0, w1, b1, wd, bd))
    assert y.shape == want_y.shape and sc.shape == want_sc.shape
    assert rel_err(y, want_y) <= 3e-6 and rel_err(sc, want_sc) <= 3e-6, (rel_err(y, want_y), rel_err(sc, want_sc))
    # channel-quad output layout [M,4,H2,W2,4]: the same values, permuted
    yq, sq = ops().stem(x, *ops().pack_stem_weights(w0, b0, w1, b1, wd, bd), quads=True)
    unq = lambda t: t.permute(0, 1, 4, 2, 3).reshape(y.shape)
    assert yq.shape == (m, 4, y.shape[2], y.shape[3], 4) and torch.equal(unq(yq), y) and torch.equal(unq(sq), sc)
    with pytest.raises(RuntimeError):
        ops().stem(torch.zeros((1, 4, 8, 8), device=DEV), *ops().pack_stem_weights(w0, b0, w1, b1, wd, bd))


@pytest.mark.parametrize("case", [(5, 256, 320), (2, 37, 70), (1, 8, 32), (3, 64, 96), (1, 5, 3)])
def test_res_chain16_one_launch_matches_torch(case, conv_arithmetic):
    """itermvs_res_chain16: FeatureNet.layer1 behind the stem (module.py:33-50, net.py:13,40) in one launch against the three
    torch layers in fp64, and against the three-launch form of this library (same bf16x3 arithmetic per layer)"""
    if conv_arithmetic != "bf16x3":
        pytest.skip("the chain exists in the bf16x3 arithmetic only (the fp32 form keeps its three launches)")
    n, h, w = case
    g = torch.Generator().manual_seed(n * 1000 + h)
    y1 = torch.randn((n, 16, h, w), generator=g).relu().to(DEV)
    sc = torch.randn((n, 16, h, w), generator=g).to(DEV)
    wts = [(torch.randn((16, 16, 3, 3), generator=g) * 0.12).to(DEV) for _ in range(3)]
    bs = [(torch.randn((16,), generator=g) * 0.2).to(DEV) for _ in range(3)]
    d = lambda t: t.double()
    a = F.relu(F.conv2d(d(y1), d(wts[0]), d(bs[0]), padding=1) + d(sc))
    b = F.relu(F.conv2d(a, d(wts[1]), d(bs[1]), padding=1))
    want = F.relu(F.conv2d(b, d(wts[2]), d(bs[2]), padding=1) + a).float()
    pk = [ops().MfmaWeight(wt, split3=True) for wt in wts]
    got = ops().res_chain16(y1, sc, pk, bs)
    assert got.shape == want.shape and rel_err(got, want) <= 3e-6, rel_err(got, want)
    a3 = ops().conv2d(y1, pk[0], bs[0], act="relu", add=sc)
    b3 = ops().conv2d(a3, pk[1], bs[1], act="relu")
    c3 = ops().conv2d(b3, pk[2], bs[2], act="relu", add=a3)
    assert rel_err(got, c3) <= 2e-6, rel_err(got, c3)
    # image strides: slices of larger buffers
    big_y, big_s, big_o = (torch.zeros((n, 20, h, w), device=DEV) for _ in range(3))
    big_y[:, 2:18] = y1
    big_s[:, 1:17] = sc
    out = ops().res_chain16(big_y[:, 2:18], big_s[:, 1:17], pk, [bs[0], None, bs[2]], out=big_o[:, 4:20])
    b_nb = F.relu(F.conv2d(a, d(wts[1]), None, padding=1))
    want_nb = F.relu(F.conv2d(b_nb, d(wts[2]), d(bs[2]), padding=1) + a).float()
    assert rel_err(out, want_nb) <= 3e-6 and float(big_o[:, :4].abs().max()) == 0.0
    # channel-quad inputs (ops.stem(..., quads=True)'s layout): bit-identical to the plane form
    toq = lambda t: t.reshape(n, 4, 4, h, w).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(ops().res_chain16(toq(y1), toq(sc), pk, bs, quads=True), got)
    with pytest.raises(RuntimeError):
        ops().res_chain16(y1, sc, pk, bs, quads=True)
    with pytest.raises(RuntimeError):
        ops().res_chain16(y1, sc, [ops().MfmaWeight(wt, split3=False) for wt in wts], bs)


@pytest.mark.parametrize("case", [(5, 256, 320), (2, 38, 70), (1, 8, 32), (3, 64, 96), (1, 2, 2), (1, 10, 34)])
@pytest.mark.parametrize("storage", ["planes", "fp32", "fp16", "bf16"])
def test_lateral_conv3x3_one_launch_matches_torch(case, storage):
    """itermvs_lateral_conv3x3 (net.py:48-50: F.interpolate x2 + inner1, then output1, the 48-channel map kept on the chip) against
    the torch layers in fp64 and against the two-launch form of this library (fused-interpolate lateral layer, then the bf16x3
    3x3 layer): full level-1 size of cfg 1, ragged sizes, images smaller than a tile; every output storage form"""
    n, h, w = case
    if storage != "planes" and case not in ((5, 256, 320), (2, 38, 70), (1, 2, 2)):
        pytest.skip("storage forms: three shapes")
    g = torch.Generator().manual_seed(n * 977 + h + w)
    fine = torch.randn((n, 16, h, w), generator=g).relu().to(DEV)
    coarse = torch.randn((n, 48, h // 2, w // 2), generator=g).to(DEV)
    wl = (torch.randn((48, 16, 1, 1), generator=g) * 0.25).to(DEV)
    bl = (torch.randn((48,), generator=g) * 0.2).to(DEV)
    wo = (torch.randn((16, 48, 3, 3), generator=g) * 0.07).to(DEV)
    bo = (torch.randn((16,), generator=g) * 0.2).to(DEV)
    d = lambda t: t.double()
    intra = F.interpolate(d(coarse), scale_factor=2, mode="bilinear") + F.conv2d(d(fine), d(wl), d(bl))
    want = F.conv2d(intra, d(wo), d(bo), padding=1).float()
    pl, po = ops().MfmaWeight(wl), ops().MfmaWeight(wo, split3=True)
    two = ops().conv2d(ops().conv2d(fine, pl, bl, ksize=1, pad=0, add=coarse, add_up2=True), po, bo)
    if storage == "planes":
        got = ops().lateral_conv3x3(fine, coarse, pl, bl, po, bo)
        assert got.shape == want.shape and got.is_contiguous()
        tol = 3e-6
    else:
        dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[storage]
        out = torch.empty((n, 16, h, w), device=DEV, dtype=dt, memory_format=torch.channels_last)
        out2 = torch.empty((n, 16, h, w), device=DEV)
        got = ops().lateral_conv3x3(fine, coarse, pl, bl, po, bo, out=out, channels_last_out=True, out2=out2)
        assert got is out and rel_err(out2, want) <= 3e-6
        assert torch.equal(out.float(), out2.to(dt).float())            # the 16-bit forms round the fp32 result to nearest even
        got, tol = out.float(), {"fp32": 3e-6, "fp16": 1e-3, "bf16": 8e-3}[storage]
    assert rel_err(got, want) <= tol, rel_err(got, want)
    if storage == "planes":
        assert rel_err(got, two) <= 2e-6, rel_err(got, two)
        # no biases; image strides: channel slices of larger buffers
        big_f, big_c, big_o = torch.zeros((n, 20, h, w), device=DEV), torch.zeros((n, 50, h // 2, w // 2), device=DEV), torch.zeros((n, 18, h, w), device=DEV)
        big_f[:, 3:19] = fine
        big_c[:, 1:49] = coarse
        o = ops().lateral_conv3x3(big_f[:, 3:19], big_c[:, 1:49], pl, None, po, None, out=big_o[:, 2:18])
        intra_nb = F.interpolate(d(coarse), scale_factor=2, mode="bilinear") + F.conv2d(d(fine), d(wl))
        assert rel_err(o, F.conv2d(intra_nb, d(wo), padding=1).float()) <= 3e-6 and float(big_o[:, :2].abs().max()) == 0.0
        with pytest.raises(RuntimeError):
            ops().lateral_conv3x3(fine, coarse[:, :, :, :-1], pl, bl, po, bo)
        with pytest.raises(RuntimeError):
            ops().lateral_conv3x3(fine, coarse, pl, bl, ops().MfmaWeight(wo, split3=False), bo)
