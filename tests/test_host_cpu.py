"""CPU-only checks: the C-ABI library loads and exports everything include/itermvs_hip.h
declares, argument validation returns the documented error codes without touching a GPU, the
state-dict schema equals the published checkpoint's, and the product path refuses CPU tensors."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT, load_weights


def test_library_exports_every_declared_symbol():
    from itermvs_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "itermvs_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(itermvs_\w+)\s*\(", header, flags=re.M))
    assert declared, "no prototypes parsed from the header"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.itermvs_version() == _lib.ABI_VERSION
    m = re.search(r"#define ITERMVS_ABI_VERSION (\d+)", header)
    assert int(m.group(1)) == _lib.ABI_VERSION


def test_library_exports_nothing_but_the_declared_symbols():
    """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table IS the header (no mangled launch helpers, no
    kernel handle variables): what a binding can see is exactly what include/itermvs_hip.h documents"""
    import subprocess
    from itermvs_amd import _lib
    so = os.path.join(ROOT, "itermvs_amd", "libitermvs_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert names == set(_lib.PROTOTYPES), sorted(names ^ set(_lib.PROTOTYPES))[:10]


def test_struct_layout_matches_header_sizes():
    """ctypes mirrors of the C structs: sizes follow from the header's field lists."""
    from itermvs_amd import _lib
    assert C.sizeof(_lib.FMap) == 8 + 4 * 8 + 4 * 4
    assert C.sizeof(_lib.LevelSrc) == 16 * 8 + 4 * 8 + 4 * 4
    assert C.sizeof(_lib.CorrIterParams) == 8 * 4 + 3 * C.sizeof(_lib.LevelSrc) + 3 * 8 + 3 * 8 + 3 * 8 + 8 + 8 + 3 * 8 * 4 + 2 * 8 + 3 * 8
    assert C.sizeof(_lib.CorrInitParams) == 6 * 4 + C.sizeof(_lib.LevelSrc) + C.sizeof(_lib.FMap) + 5 * 8
    assert C.sizeof(_lib.TapParams) == 8 * 4 + 3 * 8 + 8 + 8 * 4 + 4 * 8


def test_argument_validation_error_codes():
    """Validation happens before any launch, so these calls are safe without a GPU."""
    from itermvs_amd import _lib
    lib = _lib.load()
    assert lib.itermvs_error_string(0) == b"ok"
    assert lib.itermvs_compose_proj(None, 1, 2, None, None, None, None, 0, None, None, None) == -1            # ERR_NULL
    assert lib.itermvs_corr_iter(None, None) == -1
    buf = (C.c_float * 64)()
    addr = C.addressof(buf)
    assert lib.itermvs_compose_proj(addr, 0, 2, addr, None, None, None, 0, None, None, None) == -2            # ERR_DIMS
    assert lib.itermvs_compose_proj(addr, 1, 40, addr, None, None, None, 0, None, None, None) == -4           # ERR_VIEWS
    assert lib.itermvs_softmax_max(addr, 0, 4, 4, addr, None) == -2
    p = _lib.CorrInitParams()
    p.B, p.S, p.H, p.W, p.N = 1, 1, 4, 4, 32
    p.ref.data = addr; p.proj = addr; p.inv_depth_min = addr; p.inv_depth_max = addr; p.out = addr
    p.src.view[0] = addr
    p.src.C, p.src.H, p.src.W = 20, 4, 4
    p.src.sb, p.src.sc, p.src.sy, p.src.sx = 320, 1, 80, 20
    assert lib.itermvs_corr_init(C.byref(p), None) == -3                           # ERR_CHANNELS
    p.src.C = 48; p.src.sc = 16
    assert lib.itermvs_corr_init(C.byref(p), None) == -6                           # ERR_LAYOUT (not channels-last)
    p.src.sc = 1; p.src.sx = 50
    assert lib.itermvs_corr_init(C.byref(p), None) == -5                           # ERR_ALIGN
    p.S = 17
    assert lib.itermvs_corr_init(C.byref(p), None) == -4                           # ERR_VIEWS
    # the tap-index diagnostic
    assert lib.itermvs_tap_indices(None, None) == -1
    t = _lib.TapParams()
    t.B, t.S, t.H, t.W, t.N, t.H1, t.W1 = 1, 1, 4, 4, 4, 4, 4
    t.proj = addr; t.inv_depth_min = addr; t.inv_depth_max = addr; t.out = addr
    assert lib.itermvs_tap_indices(C.byref(t), None) == -1                         # no hypotheses source (ERR_NULL)
    t.norm_depth = addr; t.N = 9
    assert lib.itermvs_tap_indices(C.byref(t), None) == -2                         # more offsets than ITERMVS_MAX_HYP
    t.S = 17
    assert lib.itermvs_tap_indices(C.byref(t), None) == -4                         # ERR_VIEWS
    # training BatchNorm entry points: scratch size = one (count, mean, M2) triple per 8192-float slab + 3 floats per channel
    assert lib.itermvs_bn_workspace_floats(20, 8, 512 * 640) == 8 * 20 * 40 * 3 + 8 * 3
    assert lib.itermvs_bn_workspace_floats(3, 16, 33 * 20) == 16 * 3 * 1 * 3 + 16 * 3
    assert lib.itermvs_bn_workspace_floats(0, 8, 64) == -2
    assert lib.itermvs_bn_train_forward(None, addr, 1, 1, 64, addr, addr, 1e-5, 0.1, 1, None, None, addr, addr, addr, None) == -1
    assert lib.itermvs_bn_train_forward(addr, addr, 1, 1, 1, addr, addr, 1e-5, 0.1, 1, None, None, addr, addr, addr, None) == -2   # one value per channel
    assert lib.itermvs_bn_train_backward(addr, None, addr, 1, 1, 64, addr, addr, addr, addr, 1, addr, addr, addr, None) == -1
    assert lib.itermvs_bn_train_backward(addr, addr, addr, 1, 70000, 64, addr, addr, addr, addr, 1, addr, addr, addr, None) == -2   # C > 65535
    for code in range(-7, 1):
        assert len(lib.itermvs_error_string(code)) > 0


def test_schema_equals_published_checkpoint(weights_dtu):
    from itermvs_amd.schema import check_state_dict, state_dict_schema, strip_module_prefix
    check_state_dict(weights_dtu)
    assert len(state_dict_schema()) == 150
    n_params = sum(int(torch.tensor(shp).prod()) if shp else 1 for k, shp in state_dict_schema().items()
                   if "running" not in k and "num_batches" not in k)
    assert n_params == 343685
    pref = {"module." + k: v for k, v in weights_dtu.items()}
    assert set(strip_module_prefix(pref)) == set(weights_dtu)
    with pytest.raises(KeyError):
        check_state_dict({k: v for k, v in list(weights_dtu.items())[:-1]})


def test_pipeline_state_dict_names_and_cpu_rejection(weights_dtu):
    from itermvs_amd import synthetic
    from itermvs_amd.net import Pipeline
    from itermvs_amd.schema import state_dict_schema
    m = Pipeline(iteration=4, test=True)
    assert list(m.state_dict().keys()) == list(state_dict_schema().keys())
    assert sum(p.numel() for p in m.parameters()) == 343685
    m.load_checkpoint_state({"module." + k: v for k, v in weights_dtu.items()})    # eval.py:124-125 format
    assert torch.equal(m.state_dict()["iter_mvs.update.gru.convq.bias"], weights_dtu["iter_mvs.update.gru.convq.bias"])
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in list(weights_dtu.items())[:-1]})       # strict, like the reference
    s = synthetic.make_sample(batch=1, num_views=3, height=64, width=96, seed=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(s["imgs"], s["proj_matrices"], s["depth_min"], s["depth_max"])


def test_ops_refuse_cpu_tensors():
    from itermvs_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.bilinear_up(torch.zeros(1, 1, 4, 4), 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.compose_proj(torch.eye(4).repeat(1, 2, 1, 1))
    # the round-2 entry points: CPU tensors are refused before anything is launched
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.stem(torch.zeros(1, 3, 8, 8), torch.zeros(ops.STEM_W0_FLOATS), torch.zeros(ops.STEM_W1_FLOATS))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.corrnet(torch.zeros(1, 8, 8, 8), [torch.zeros(ops.CORRNET_WEIGHT_FLOATS)])
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.copy_multi([torch.zeros(4)], [torch.ones(4)])
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.view_aggregate_up(torch.zeros(1, 2, 4, 8, 4, 4), torch.zeros(1, 2, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.final_upsample(torch.zeros(1, 144, 4, 4), torch.zeros(1, 43, 4, 4), torch.ones(1), torch.ones(1), torch.zeros(1, 1, 4, 4))


def test_weight_packings_of_the_fused_kernels():
    """pack_stem_weights / pack_corrnet_weights lay the weights out as include/itermvs_hip.h documents (CPU, no launch)"""
    from itermvs_amd import ops
    g = torch.Generator().manual_seed(2)
    w0, b0 = torch.randn((8, 3, 3, 3), generator=g), torch.randn((8,), generator=g)
    w1, b1 = torch.randn((16, 8, 3, 3), generator=g), torch.randn((16,), generator=g)
    wd, bd = torch.randn((16, 8, 3, 3), generator=g), torch.randn((16,), generator=g)
    p0, p1 = ops.pack_stem_weights(w0, b0, w1, b1, wd, bd)
    assert p0.numel() == ops.STEM_W0_FLOATS and p1.numel() == ops.STEM_W1_FLOATS
    ci, ky, kx, co = 2, 1, 2, 5
    assert p0[((ci * 3 + ky) * 3 + kx) * 8 + co] == w0[co, ci, ky, kx] and torch.equal(p0[216:], b0)
    tap, ks, q = ky * 3 + kx, 1, 3                       # ci = 4 * ks + q = 7
    assert p1[((tap * 2 + ks) * 4 + q) * 32 + co] == w1[co, 7, ky, kx]
    assert p1[((tap * 2 + ks) * 4 + q) * 32 + 16 + co] == wd[co, 7, ky, kx]
    assert torch.equal(p1[2304:2320], b1) and torch.equal(p1[2320:], bd)
    with pytest.raises(RuntimeError):
        ops.pack_stem_weights(w0[:4], b0, w1, b1, wd, bd)
    # CorrNet: conv0 in its two-rows-per-tile form
    names = {"conv0.conv.weight": (8, 8, 3, 3), "conv1.conv.weight": (16, 8, 3, 3), "conv2.conv.weight": (32, 16, 3, 3),
             "conv3.weight": (32, 16, 3, 3), "conv4.weight": (16, 8, 3, 3), "conv5.weight": (1, 8, 3, 3), "conv5.bias": (1,)}
    w = {"p." + k: torch.randn(v, generator=g) for k, v in names.items()}
    pk = ops.pack_corrnet_weights(w, "p.")
    assert pk.numel() == ops.CORRNET_WEIGHT_FLOATS
    pk3 = ops.pack_corrnet_weights(w, "p.", split3=True)
    assert pk3.numel() == ops.CORRNET_WEIGHT_FLOATS_SPLIT3 == 23120
    assert torch.equal(pk3[4608:5760], pk[1536:2688]) and torch.equal(pk3[23040:], pk[14208:])      # conv1, conv5 + bias unchanged
    # conv2 and the two transposed convolutions: weight_format 3 = bf16 [tap][chunk][h, m, l][row co][16 ci]; the three terms add up
    # to the fp32 weight exactly; a transposed convolution is packed as the convolution weight [co][ci][ky][kx] = w[ci][co][ky][kx]
    for off, rows, chunks, wt in ((5760, 32, 1, w["p.conv2.conv.weight"]), (12672, 16, 2, w["p.conv3.weight"].permute(1, 0, 2, 3)),
                                  (19584, 16, 1, w["p.conv4.weight"].permute(1, 0, 2, 3))):
        n = 9 * chunks * 3 * rows * 16 // 2
        t = pk3[off:off + n].view(torch.int16).view(torch.bfloat16).reshape(9, chunks, 3, rows, 16).float().sum(2)     # [tap, chunk, co, ci16]
        want = torch.zeros((9, chunks, rows, 16))
        co, ci = wt.shape[0], wt.shape[1]
        want.view(9, chunks, rows, 16)[:, :, :co, :] = wt.float().permute(2, 3, 0, 1).reshape(9, co, chunks, 16).permute(0, 2, 1, 3)
        assert torch.equal(t, want), off
    # conv0's bf16 operands: MFMA 2v / 2v+1 / 12+v of window-position pair v hold terms [h h|h h] / [m m|m m] / [l l|h h]; their sum over
    # the three terms of a position is the fp32 weight exactly
    a = pk3[:4608].view(torch.int16).view(torch.bfloat16).reshape(18, 4, 16, 8).float()
    w0c = w["p.conv0.conv.weight"].float()
    two = torch.zeros((4, 3, 8, 16))
    two[0:3, :, :, 0:8] = w0c.permute(2, 3, 1, 0)
    two[1:4, :, :, 8:16] = w0c.permute(2, 3, 1, 0)
    full = two.reshape(12, 8, 16).permute(0, 2, 1)                                   # [window position, row, ci]
    for v in range(6):
        for sel in (0, 1):
            assert torch.equal(a[2 * v, sel] + a[2 * v + 1, sel] + a[12 + v, sel], full[2 * v + sel])
            assert torch.equal(a[2 * v, sel], a[2 * v, 2 + sel]) and torch.equal(a[12 + v, 2 + sel], a[2 * v, sel])
    c0 = w["p.conv0.conv.weight"]
    wr, kx, ks, q, co = 2, 1, 1, 2, 3                    # window row 2: tap row 2 of the first output row, tap row 1 of the second
    idx = (((wr * 3 + kx) * 2 + ks) * 4 + q) * 16
    assert pk[idx + co] == c0[co, 4 * ks + q, 2, kx] and pk[idx + 8 + co] == c0[co, 4 * ks + q, 1, kx]
    idx0 = (((0 * 3 + kx) * 2 + ks) * 4 + q) * 16       # window row 0 is outside the second output row's taps
    assert pk[idx0 + co] == c0[co, 4 * ks + q, 0, kx] and pk[idx0 + 8 + co] == 0


def test_gru_conv_weight_packing():
    """pack_gru_conv_split3 lays the ConvGRU weights out as include/itermvs_hip.h documents for itermvs_gru_conv (CPU, no launch):
    three exact bf16 terms per weight; channels 0..31 as one term per operand, channels 32..42 as the two-term operands
    A1 = [h | h], A2 = [m | m], A3 = [l | h]"""
    from itermvs_amd import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn((64, 43, 3, 3), generator=g)
    wp = ops.pack_gru_conv_split3(w)
    assert wp.dtype == torch.bfloat16 and tuple(wp.shape) == (4, 9, 6, 64, 8)
    f = wp.float()
    back = f[:, :, :3].sum(2).reshape(4, 9, 4, 16, 2, 4).permute(0, 3, 4, 2, 5, 1).reshape(64, 32, 3, 3)       # (ob, i), (jj, q, r), tap
    assert torch.equal(back, w[:, :32])
    ob, tap, q, i, j = 2, 5, 3, 7, 6                     # lane 16 q + i, slot j: channel (j // 4) * 16 + 4 q + j % 4
    c = (j // 4) * 16 + 4 * q + j % 4
    assert float(f[ob, tap, :3, 16 * q + i, j].sum()) == float(w[16 * ob + i, c, tap // 3, tap % 3])
    b = f[:, :, 3:].reshape(4, 9, 3, 2, 2, 16, 8)        # [ob, tap, operand, second, half, i, j]
    assert torch.equal(b[:, :, 0, 0], b[:, :, 0, 1]) and torch.equal(b[:, :, 1, 0], b[:, :, 1, 1])             # A1 = [h | h], A2 = [m | m]
    assert torch.equal(b[:, :, 2, 1], b[:, :, 0, 0])                                                            # A3 = [l | h]
    tail = (b[:, :, 0, 0] + b[:, :, 1, 0] + b[:, :, 2, 0]).permute(0, 3, 2, 4, 1).reshape(64, 16, 3, 3)         # (ob, i), (half, j), tap
    assert torch.equal(tail[:, :11], w[:, 32:]) and float(tail[:, 11:].abs().max()) == 0.0
    assert tuple(ops.pack_gru_conv_split3(w[:32]).shape) == (2, 9, 6, 64, 8)
    with pytest.raises(RuntimeError):
        ops.pack_gru_conv_split3(w[:, :40])
    with pytest.raises(RuntimeError):
        ops.pack_gru_conv_split3(w[:24])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "itermvs_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle under ``oracle/``", ""), fn


def test_synthetic_inputs_are_deterministic_and_dtu_shaped():
    from itermvs_amd import synthetic
    a = synthetic.make_sample(1, 5, 64, 96, seed=3)
    b = synthetic.make_sample(1, 5, 64, 96, seed=3)
    assert torch.equal(a["imgs"]["level_0"], b["imgs"]["level_0"])
    assert a["imgs"]["level_2"].shape == (1, 5, 3, 16, 24)
    p0, p1 = a["proj_matrices"]["level_0"][0, 0], a["proj_matrices"]["level_1"][0, 0]
    assert torch.allclose(p1[:2], p0[:2] / 2) and torch.equal(p1[2:], p0[2:])     # K rows 0-1 halve per level
    assert torch.equal(p0[3], torch.tensor([0.0, 0.0, 0.0, 1.0]))
    w0 = synthetic.random_state_dict(0)
    assert all(torch.equal(w0[k], v) for k, v in load_weights("seed0").items())   # matches the golden weights
    sc = synthetic.make_scene_sample(3, 64, 96, seed=1)
    assert 425 < float(sc["depth_gt"].min()) and float(sc["depth_gt"].max()) < 935


def test_bench_algorithmic_bytes_match_design():
    import bench
    it, init, per_map = bench.algorithmic_bytes(4, 512, 640, 1, 4)
    assert abs(it / 1e6 - 50.2) < 0.3 and abs(per_map / 1e6 - 229.0) < 3.0


def test_weight_packings_follow_the_documented_layouts():
    """host-side re-layouts of the convolution / head weights (include/itermvs_hip.h), element by element"""
    import torch
    from itermvs_amd import ops
    gen = torch.Generator().manual_seed(0)
    for cout, cin in ((16, 3), (16, 8), (32, 43), (48, 48)):
        w = torch.randn((cout, cin, 3, 3), generator=gen)
        pk = ops.MfmaWeight(w)
        s = ops.tile_k_split(cin)
        nch = (cin + 4 * s - 1) // (4 * s)
        cp = (cout + 15) // 16 * 16
        assert pk.data.shape == (9, (cin + 3) // 4 * 4, cp) and pk.tile.shape == (9, nch, 4, cp, s)
        for tap, ch, q, co, k in ((0, 0, 0, 0, 0), (4, nch - 1, 3, cout - 1, s - 1), (8, 0, 2, 5, 0), (7, nch // 2, 1, cp - 1, s - 1)):
            c = ch * 4 * s + q * s + k
            want = float(w[co, c, tap // 3, tap % 3]) if (c < cin and co < cout) else 0.0
            assert float(pk.tile[tap, ch, q, co, k]) == want
            assert float(pk.data[tap, min(c, pk.data.shape[1] - 1), co]) == (want if c < pk.data.shape[1] else float(pk.data[tap, -1, co]))
    wt = torch.randn((32, 16, 3, 3), generator=gen)                    # ConvTranspose2d weight [Cin, Cout, k, k]
    pt = ops.MfmaWeight(wt, transposed=True)
    assert pt.transposed and pt.cin == 32 and pt.cout == 16
    assert float(pt.tile[5, 1, 2, 7, 3]) == float(wt[1 * 16 + 2 * 4 + 3, 7, 1, 2])
    w1, w2 = torch.randn((64, 32, 1, 1), generator=gen), torch.randn((256, 64, 1, 1), generator=gen)
    a1, a2 = ops.pack_head_weights(w1, w2)
    assert a1.shape == (4, 2, 4, 16, 4) and a2.shape == (16, 4, 4, 16, 4)
    assert float(a1[3, 1, 2, 5, 3]) == float(w1[3 * 16 + 5, 1 * 16 + 2 * 4 + 3, 0, 0])
    assert float(a2[9, 2, 1, 15, 0]) == float(w2[9 * 16 + 15, 2 * 16 + 1 * 4 + 0, 0, 0])
    v = ops.pack_conv_weight(w)                                        # VALU format [Cin, k, k, Cout]
    assert v.shape == (48, 3, 3, 48) and float(v[7, 2, 1, 40]) == float(w[40, 7, 2, 1])


def test_bench_self_launch_and_workload_names():
    """`python bench.py --gpus 8` without a launcher re-executes itself under torch.distributed.run (one rank per GPU,
    rendezvous on 127.0.0.1); the bench line names the BASELINE config that actually ran"""
    import sys
    import bench
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(bench.os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert bench.workload_name(5, 512, 640, 4).startswith("BASELINE cfg 2")
    assert bench.workload_name(5, 1152, 1600, 4).startswith("BASELINE cfg 3")
    assert bench.workload_name(11, 1280, 1920, 8).startswith("BASELINE cfg 5")
    assert bench.workload_name(3, 64, 96, 2).startswith("custom")


def test_timed_regions_and_median():
    from itermvs_amd import shard
    calls = []
    regions = shard.timed_regions(lambda i: calls.append(i), steps=4, warmup=3, repeats=5)
    assert len(regions) == 5 and all(r >= 0 for r in regions)
    assert calls == list(range(3 + 4 * 5))                       # a running step index: warm-up, then 5 regions of 4
    assert shard.median([3.0, 1.0, 2.0]) == 2.0 and shard.median([4.0, 1.0, 2.0, 3.0]) == 2.5


def test_feature_grad_pool_matches_per_call_gradients():
    """ops.FeatureGradPool / feature_grad_sink (training: one shared accumulator per pyramid level for all Evaluation calls
    of a step): with stand-in Functions on the CPU the pooled form must give the gradients of the per-call form, the sink's
    backward must run after every pooled call's, and other consumers' gradients must be added"""
    from itermvs_amd.ops import FeatureGradPool, feature_grad_sink
    order = []

    class Call(torch.autograd.Function):              # stand-in for _CorrIterFn: y = k * sum(f_l); pooled backward scatters
        @staticmethod
        def forward(ctx, k, f1, f2, f3, pool):
            ctx.k, ctx.pool = k, pool
            ctx.save_for_backward(f1, f2, f3)
            return k * (f1.sum() + 2 * f2.sum() + 3 * f3.sum())

        @staticmethod
        def backward(ctx, g):
            order.append(("call", ctx.k))
            grads = []
            for l, f in zip((1, 2, 3), ctx.saved_tensors):
                contrib = torch.full_like(f, float(g) * ctx.k * l)
                if ctx.pool is None:
                    grads.append(contrib)
                else:
                    ctx.pool.get(l, f).add_(contrib)
                    grads.append(None)
            return (None,) + tuple(grads) + (None,)

    def run(pooled):
        torch.manual_seed(0)
        leaves = {l: torch.randn(2, 3, 4, 5).requires_grad_(True) for l in (1, 2, 3)}
        feats = {l: leaves[l] * 1.5 for l in leaves}
        pool = FeatureGradPool() if pooled else None
        if pooled:
            feats = feature_grad_sink(pool, feats)
        loss = sum(Call.apply(float(k), feats[1], feats[2], feats[3], pool) for k in (1, 2, 3))
        loss = loss + (feats[2] ** 2).sum()                        # another consumer of a level (like the up-sampling head)
        loss.backward()
        assert pool is None or not pool.acc                         # handed over and released
        return {l: leaves[l].grad.clone() for l in leaves}

    plain = run(False)
    order.clear()
    pooled = run(True)
    assert len(order) == 3
    for l in (1, 2, 3):
        assert torch.allclose(plain[l], pooled[l], rtol=1e-6, atol=1e-6), l


def test_library_load_brings_torch_in_first():
    """_lib.load() in a fresh interpreter: torch (and with it PyTorch-ROCm's bundled HIP runtime) is imported before the
    dlopen of libitermvs_hip.so -- see tests/test_drivers_gpu.py::test_library_loaded_before_torch_then_smoke for why"""
    import subprocess
    import sys
    code = ("import sys, ctypes; real = ctypes.CDLL\n"
            "def spy(path, *a, **k):\n"
            "    if 'libitermvs_hip' in str(path): print('TORCH_FIRST', 'torch' in sys.modules)\n"
            "    return real(path, *a, **k)\n"
            "ctypes.CDLL = spy\n"
            "from itermvs_amd import _lib; _lib.load()")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "TORCH_FIRST True" in p.stdout, (p.stdout, p.stderr[-800:])


def test_algorithmic_bytes_are_the_designs_figures():
    """bench.py's roofline numerators (SURVEY 8(d) with hypotheses built in-kernel): DESIGN.md section 4 quotes these figures"""
    from itermvs_amd.benchmarks import algorithmic_bytes
    it, init, per_map = algorithmic_bytes(4, 512, 640, 1, 4)                       # cfg 1, fp32
    assert round(it / 1e6, 1) == 50.2 and round(init / 1e6, 1) == 25.9 and round(per_map / 1e6, 1) == 226.8
    it16, init16, _ = algorithmic_bytes(4, 512, 640, 1, 4, e=2)                    # 16-bit feature storage
    assert round(it16 / 1e6, 1) == 32.5 and init16 < init
    it5, _, _ = algorithmic_bytes(10, 1280, 1920, 1, 8)                            # cfg-5 shape, fp32
    assert 770e6 < it5 < 790e6


def test_split_bf16x3_is_exact_and_conv_arithmetic_is_validated():
    """the host side of the bf16x3 convolutions: h + m + l reproduces every finite fp32 value exactly (three bf16 terms), and
    the engine rejects unknown arithmetic names before touching the GPU"""
    from itermvs_amd import ops
    from itermvs_amd.engine import InferenceEngine
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4096, generator=g) * s for s in (1e-20, 1e-3, 1.0, 1e6)] + [torch.tensor([0.0, -0.0, 1.0, -1.5, 3.4e38, 1e-30])])      # (below ~1e-33 the low terms fall under bf16's smallest subnormal: absolute error < 1e-40)
    h, m, l = ops.split_bf16x3(x)
    assert h.dtype == m.dtype == l.dtype == torch.bfloat16
    back = h.double() + m.double() + l.double()
    assert torch.equal(back, x.double())
    with pytest.raises(ValueError):
        InferenceEngine({}, 4, conv_arithmetic="tf32")


def test_build_rejects_unknown_and_knockout_switches():
    """the product build must not be reachable with a results-corrupting knock-out macro or a mistyped ITERMVS_* switch
    (VERDICT r05 item 8): `make` stops on them unless TUNING=1 (knock-outs) / always (unknown names); the product sources hold
    no knock-out code at all (it lives as patches under csrc/experiments/)."""
    import os, subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "itermvs_amd", "csrc")
    run = lambda *a: subprocess.run(["make", "-n", "-C", csrc] + list(a), capture_output=True, text=True)  # noqa: E731
    assert run().returncode == 0
    for bad in ("EXTRA=-DITERMVS_TILE3_KO=8", "EXTRA=-DITERMVS_TILE_KO_BF16=3", "EXTRA=-DITERMVS_NO_SUCH_SWITCH", "CXXFLAGS=-DITERMVS_HEAD_KO=1"):
        r = run(bad)
        assert r.returncode != 0 and "unknown or forbidden build switch" in r.stderr, (bad, r.stderr[-300:])
    assert run("TUNING=1", "EXTRA=-DITERMVS_TILE3_KO=8").returncode == 0
    assert run("EXTRA=-DITERMVS_EXACT_DIV").returncode == 0
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")) and f != "common.hpp":
            text = open(os.path.join(csrc, f)).read()
            assert "_KO" not in text, f"knock-out code in product source {f}"


def test_rank_binds_to_the_cpus_of_its_gpus_numa_node(tmp_path):
    """shard.bind_to_gpu_numa: PCI address -> sysfs numa_node / local_cpulist -> sched_setaffinity,
    intersected with the mask the process already has; no node / empty intersection / ITERMVS_NO_AFFINITY leave it alone."""
    import os, subprocess, sys, textwrap
    from itermvs_amd import shard
    assert shard.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert shard.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11" and shard.format_cpulist([5]) == "5"
    have = sorted(os.sched_getaffinity(0))
    root = tmp_path / "sys"
    dev = root / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    local = have[: max(1, len(have) // 2)]
    (dev / "local_cpulist").write_text(shard.format_cpulist(local + [4093, 4094]) + "\n")        # two CPUs this process may not use
    node, cpus = shard.pci_numa_cpus("0000:C1:00.0", str(root))
    assert node == 1 and cpus == sorted(local + [4093, 4094])
    info = shard.bind_to_gpu_numa(0, str(root), bdf="0000:c1:00.0", apply=False)
    assert info == {"gpu": "0000:c1:00.0", "numa_node": 1, "cpus": shard.format_cpulist(local), "n_cpus": len(local), "bound": False}
    # node -1 (platform says nothing), unknown device, cpulist from the node directory
    dev2 = root / "bus" / "pci" / "devices" / "0000:05:00.0"
    dev2.mkdir(parents=True)
    (dev2 / "numa_node").write_text("-1\n")
    assert shard.pci_numa_cpus("0000:05:00.0", str(root)) == (None, [])
    assert shard.bind_to_gpu_numa(0, str(root), bdf="0000:05:00.0")["bound"] is False
    assert shard.bind_to_gpu_numa(0, str(root), bdf="0000:99:00.0")["bound"] is False
    dev3 = root / "bus" / "pci" / "devices" / "0000:06:00.0"
    dev3.mkdir(parents=True)
    (dev3 / "numa_node").write_text("0\n")
    nd = root / "devices" / "system" / "node" / "node0"
    nd.mkdir(parents=True)
    (nd / "cpulist").write_text(shard.format_cpulist(have[-1:]))
    assert shard.pci_numa_cpus("0000:06:00.0", str(root)) == (0, have[-1:])
    # the real thing, in a child (the mask of the test process stays as it is): bound, every later thread inherits it, and
    # restore_affinity widens all threads again
    code = textwrap.dedent(f"""
        import os, sys, threading
        sys.path.insert(0, {str(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))!r})
        from itermvs_amd import shard
        before = sorted(os.sched_getaffinity(0))
        info = shard.bind_to_gpu_numa(0, {str(root)!r}, bdf="0000:c1:00.0")
        assert info["bound"] and sorted(os.sched_getaffinity(0)) == {local!r}, (info, os.sched_getaffinity(0))
        seen = []
        t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0)))); t.start(); t.join()
        assert seen[0] == {local!r}
        assert shard.restore_affinity() and sorted(os.sched_getaffinity(0)) == before
        os.environ["ITERMVS_NO_AFFINITY"] = "1"
        assert shard.bind_to_gpu_numa(0, {str(root)!r}, bdf="0000:c1:00.0")["bound"] is False
        print("child ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stderr[-800:]
