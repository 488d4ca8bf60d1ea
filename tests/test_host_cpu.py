"""CPU: host-side mirror (state_dict layout, config), C-ABI library exports, no compute."""
import ctypes
import os
import re

import pytest
import torch

import kbnet_amd as kb
from conftest import ROOT, load_golden


def _model(cfg):
    return kb.modules.KBNetModel.from_config(cfg, device=torch.device("cpu"))


@pytest.mark.parametrize("preset", ["kitti", "void"])
def test_state_dict_layout_matches_reference_keys(preset):
    cfg = kb.PRESETS[preset]()
    m = _model(cfg.narrow())
    ncfg = cfg.narrow()
    for mod, shapes in zip(m.modules(), (kb.config.s2d_param_shapes(ncfg), kb.config.encoder_param_shapes(ncfg),
                                         kb.config.decoder_param_shapes(ncfg))):
        sd = mod.state_dict()
        assert set(sd.keys()) == set(shapes.keys())
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), k


def test_full_model_parameter_count():
    m = _model(kb.kitti_config())
    assert sum(p.numel() for p in m.parameters()) == 6957780


def test_golden_state_dicts_load_with_and_without_module_prefix():
    g = load_golden("fwd_kitti")
    m = _model(kb.kitti_config().narrow())
    m.load_state_dicts(g["s2d"], g["encoder"], g["decoder"])
    pref = lambda d: {"module." + k: v for k, v in d.items()}
    m.load_state_dicts(pref(g["s2d"]), pref(g["encoder"]), pref(g["decoder"]))
    w = m.encoder.calibrated_backprojection2.conv_fused.conv.weight
    assert torch.equal(w, g["encoder"]["calibrated_backprojection2.conv_fused.conv.weight"])


def test_checkpoint_roundtrip(tmp_path):
    cfg = kb.void_config().narrow()
    a, b = _model(cfg), _model(cfg)
    path = os.path.join(tmp_path, "ckpt.pth")
    a.save_model(path, step=7)
    ckpt = torch.load(path)
    assert all(k.startswith("module.") for k in ckpt["encoder_state_dict"])
    step, _ = b.restore_model(path)
    assert step == 7
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)


def test_reference_written_checkpoint_loads():
    """f3: a .pth written by the reference's own KBNetModel.save_model (tests/golden/gen_golden.py gen_checkpoint,
    reference src/kbnet_model.py:353-376): module.-prefixed keys, three state_dicts, optimizer state, train_step."""
    from conftest import GOLDEN_DIR
    g = load_golden("fwd_kitti")
    m = _model(kb.kitti_config().narrow())
    step, _ = m.restore_model(os.path.join(GOLDEN_DIR, "ckpt_kitti_narrow.pth"))
    assert step == 1234
    for mod, name in zip(m.modules(), ("s2d", "encoder", "decoder")):
        sd = mod.state_dict()
        assert set(sd) == set(g[name])
        for k, v in sd.items():
            assert torch.equal(v, g[name][k]), k


def test_kb_layer_at_resolution_4_keeps_the_reference_state_dict():
    """4 in resolutions_backprojection: the reference builds calibrated_backprojection5 (and no conv5_*), then calls
    calibrated_backprojection4 twice (src/networks.py:266-283, :499-517, quirk Q3).  Same keys / shapes as the golden captured from
    the reference's own model, and as config.encoder_param_shapes; the stacked-conv KB block keeps the reference's keys too."""
    import dataclasses
    g = load_golden("fwd_kb01234")
    cfg = dataclasses.replace(kb.kitti_config().narrow(), resolutions_backprojection=(0, 1, 2, 3, 4),
                              n_filters_encoder_image=(8, 16, 32, 32, 32), n_filters_encoder_depth=(4, 8, 16, 16, 16))
    m = _model(cfg)
    sd = m.encoder.state_dict()
    assert set(sd) == set(g["encoder"]) == set(kb.config.encoder_param_shapes(cfg))
    assert all(tuple(sd[k].shape) == tuple(g["encoder"][k].shape) for k in sd)
    assert any(k.startswith("calibrated_backprojection5.") for k in sd) and not any(k.startswith("conv5_") for k in sd)
    m.load_state_dicts(g["s2d"], g["encoder"], g["decoder"])
    ks = load_golden("kb_stacked")["weights"]
    blk = kb.modules.CalibratedBackprojectionBlock(8, 4, 16, 16, 8, 16, 2, 3, 2, "xavier_normal", torch.nn.LeakyReLU(0.2))
    assert blk.stacked and set(blk.state_dict()) == set(ks)
    assert all(tuple(v.shape) == tuple(ks[k].shape) for k, v in blk.state_dict().items())


def test_transpose_decoder_keeps_the_reference_state_dict():
    """deconv_type='transpose' (run_kbnet.py --deconv_type): every decoder block's up-sampling layer is a TransposeConv2d whose
    parameter is `deconvN.deconv.deconv.weight`, in x out x 3 x 3 (reference src/net_utils.py:383-390, :1416-1424).  Same keys /
    shapes as the golden captured from the reference's own model and as config.decoder_param_shapes; the golden loads strictly."""
    import dataclasses
    g = load_golden("fwd_transpose")
    cfg = dataclasses.replace(kb.kitti_config().narrow(), deconv_type="transpose")
    m = _model(cfg)
    sd = m.decoder.state_dict()
    assert set(sd) == set(g["decoder"]) == set(kb.config.decoder_param_shapes(cfg))
    assert all(tuple(sd[k].shape) == tuple(g["decoder"][k].shape) for k in sd)
    assert tuple(sd["deconv4.deconv.deconv.weight"].shape) == (cfg.n_filters_encoder_image[-1] + cfg.n_filters_encoder_depth[-1],
                                                              cfg.n_filters_decoder[0], 3, 3)
    m.load_state_dicts(g["s2d"], g["encoder"], g["decoder"])
    with pytest.raises(ValueError):
        _model(dataclasses.replace(cfg, deconv_type="bilinear"))


def test_error_behaviour_mirrors_reference():
    with pytest.raises(ValueError):  # reference src/net_utils.py:45
        kb.modules.activation_func("swish")
    with pytest.raises(ValueError):  # reference src/net_utils.py:105
        kb.modules.Conv2d(3, 8, weight_initializer="nope")
    with pytest.raises(AssertionError):  # reference src/networks.py:70-75
        kb.modules.KBNetEncoder(n_filters_image=[8, 16, 32])


def test_model_object_has_the_methods_the_reference_run_loop_calls():
    """reference src/kbnet.py:804-807, 915 (restore_model, eval, parameters, forward) and src/kbnet_model.py:140-141 (data_parallel, to)."""
    m = _model(kb.kitti_config().narrow())
    assert m.data_parallel() is m
    m.eval()
    m.to(torch.device("cpu"))
    assert sum(p.numel() for p in m.parameters()) == sum(p.numel() for mod in m.modules() for p in mod.parameters())
    for name in ("forward", "restore_model", "save_model", "parameters", "eval", "to", "data_parallel"):
        assert callable(getattr(m, name))
    with pytest.raises(RuntimeError):
        m.train()
    with pytest.raises(RuntimeError):
        m.compute_loss()


def test_hip_path_rejects_cpu_tensors():
    """No silent CPU fallback: CPU tensors fail loudly."""
    m = _model(kb.kitti_config().narrow())
    image, sparse, valid, k = kb.synthetic.make_frames(1, 32, 64)
    with pytest.raises(RuntimeError):
        m.forward(image, sparse, valid, k)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "kbnet_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(kbn_[a-z0-9_]+)\s*\(", header))
    assert {"kbn_s2d_forward", "kbn_conv2d_forward", "kbn_kb_block_forward", "kbn_depth_head_forward"} <= declared
    lib_path = kb._lib.LIB_PATH
    if not os.path.exists(lib_path):
        kb._build.build(verbose=False)
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(kb._lib.SIGNATURES.keys())
    loaded = kb._lib.load()
    assert loaded.kbn_version() == kb._lib.ABI_VERSION
    assert loaded.kbn_status_string(-2) == b"configuration outside the kernel limits"
    # pure host arithmetic, no GPU needed
    assert loaded.kbn_conv2d_packed_weight_bytes(48, 3, 3, 1) == 4 * 1 * 4 * 9 * 48
    assert loaded.kbn_conv2d_packed_weight_bytes(12, 64, 3, 1) == 4 * 1 * 64 * 9 * 16
    assert loaded.kbn_conv2d_packed_weight_bytes(96, 99, 1, 2) == 4 * 2 * 112 * 1 * 48
    assert loaded.kbn_conv2d_packed_weight_bytes(16, 19, 3, 2) == 4 * 1 * 20 * 9 * 16  # stride 2: 4-channel chunks
    assert loaded.kbn_conv2d_packed_weight_bytes(5, 5, 5, 1) == 0
    # wide 3x3 stride-1 convs carry the Winograd-domain weights (16 per channel pair) behind the direct ones
    assert loaded.kbn_conv2d_packed_weight_bytes(64, 128, 3, 1) == 4 * (128 * 9 * 64 + 128 * 16 * 64)
    assert loaded.kbn_conv2d_packed_weight_bytes(40, 96, 3, 1) == 4 * (96 * 9 * 48 + 96 * 16 * 64)
    assert loaded.kbn_conv2d_packed_weight_bytes(40, 104, 3, 1) == 4 * 104 * 9 * 48  # odd number of 8-channel chunks: direct only
    # tile choice: big maps keep the largest tile, small maps shrink it so every CU gets work
    big = kb.ops.conv_plan(8, 96, 96, 3, 2, 176, 608)
    small = kb.ops.conv_plan(8, 384, 384, 3, 2, 22, 76)
    assert (big["NB"], big["MW"]) == (3, 4) and small["MW"] < 4 and small["workgroups"] >= 256
    # Winograd regions: 4 x 16 tiles on big maps; 6 x 10 on the 22 x 76 map fills exactly one round of 256 CUs
    wide = kb.ops.conv_plan(8, 64, 128, 3, 1, 176, 608)
    assert (wide["kernel"], wide["MW"], wide["TWB"], wide["workgroups"]) == ("wino", 4, 16, 3344)
    deep = kb.ops.conv_plan(8, 256, 768, 3, 1, 22, 76)
    assert (deep["kernel"], deep["MW"], deep["TWB"], deep["workgroups"]) == ("wino", 6, 10, 256)


def test_every_entry_point_rejects_null_arguments():
    """tests/abi_null_args.py in a child process: all-null / all-zero arguments to every declared entry point -> an error status
    (KBN_ERR_*), never a crash, never KBN_OK."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "abi_null_args.py")],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("accepted all-zero arguments: []"), (res.returncode, res.stdout[-400:], res.stderr[-600:])


def test_synthetic_frames_are_deterministic_and_well_formed():
    a = kb.synthetic.make_frames(2, 32, 48, "kitti", seed=3)
    b = kb.synthetic.make_frames(2, 32, 48, "kitti", seed=3)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    image, sparse, valid, k = a
    assert image.min() >= 0 and image.max() < 1
    assert torch.equal(valid, (sparse > 0).float())
    assert torch.equal(sparse * 256, torch.round(sparse * 256))  # 16-bit PNG / 256 quantisation
    assert k[0, 2, 2] == 1 and k[0, 0, 0] > 0


def test_paired_planes_recognises_only_the_two_planes_of_one_buffer():
    """modules.paired_planes: the N x 2 x H x W view KBNetModel.forward hands to S2D instead of torch.cat([sparse, validity])."""
    import torch
    from kbnet_amd import modules
    sd, vm = modules.new_depth_input_pair(3, 5, 7, "cpu")
    sd.copy_(torch.arange(3 * 35, dtype=torch.float32).view(3, 1, 5, 7))
    vm.copy_(-torch.arange(3 * 35, dtype=torch.float32).view(3, 1, 5, 7))
    pair = modules.paired_planes(sd, vm)
    assert pair is not None and pair.is_contiguous() and torch.equal(pair, torch.cat([sd, vm], 1))
    assert pair.data_ptr() == sd.data_ptr()
    sub = modules.paired_planes(sd[1:3], vm[1:3])              # a sub-batch (the branches of a captured graph)
    assert sub is not None and torch.equal(sub, torch.cat([sd[1:3], vm[1:3]], 1))
    assert modules.paired_planes(vm, sd) is None               # wrong order
    assert modules.paired_planes(sd.clone(), vm) is None       # different buffers
    assert modules.paired_planes(sd, vm.clone()) is None
    flat = torch.zeros(2 * 35)                                 # adjacent in memory but not interleaved frame by frame
    a, b = flat[:35].view(1, 1, 5, 7), flat[35:].view(1, 1, 5, 7)
    assert torch.equal(modules.paired_planes(a, b), torch.cat([a, b], 1))   # one frame: adjacency is all it takes
    two = torch.zeros(4 * 35)
    a2, b2 = two[:70].view(2, 1, 5, 7), two[70:].view(2, 1, 5, 7)
    assert modules.paired_planes(a2, b2) is None               # two frames each, back to back: not the paired layout
    assert modules.paired_planes(sd.double(), vm.double()) is None


def test_host_switches_follow_the_library_reading_of_the_environment(monkeypatch):
    """the host mirror's A/B switches (KBN_NO_PAIR*, KBN_NO_OVERLAP, KBN_DEPTH_FRONT_FUSION ...) used to be read from
    os.environ at import / construction time, while the C side's switches follow kbn_reload_env().  They now ask the library
    (kbn_knob): one reading of the environment for both sides, refreshed together; an attribute assignment still overrides."""
    enc = kb.modules.KBNetEncoder(3, 8, resolutions_backprojection=[0, 1, 2, 3], weight_initializer="xavier_normal")
    dec = kb.modules.MultiScaleDecoder(512, 1, 1, [256, 128, 128, 64, 12], [512, 256, 128, 64, 0], "xavier_normal")
    for name in ("KBN_NO_PAIR", "KBN_NO_PAIR_ENC", "KBN_NO_PAIR_MID", "KBN_NO_PAIR_TAIL", "KBN_DEPTH_FRONT_FUSION", "KBN_NO_DEPTH_FRONT_FUSION",
                 "KBN_NO_SPLIT", "KBN_FP16_ONE_TERM"):
        monkeypatch.delenv(name, raising=False)
    kb.ops.reload_env()
    try:
        assert enc.pair_chain and dec.pair_chain and dec.pair_tail and dec.deconv4.pair_mid and kb.ops.split_enabled()
        assert not enc.fuse_s2d, "the S2D-in-the-depth-front fusion is opt-in"
        monkeypatch.setenv("KBN_NO_PAIR", "1")
        assert enc.pair_chain, "the environment alone changes nothing: the library read it at load time"
        kb.ops.reload_env()
        assert not enc.pair_chain and not dec.pair_chain and dec.pair_tail
        monkeypatch.delenv("KBN_NO_PAIR")
        monkeypatch.setenv("KBN_NO_PAIR_ENC", "1")
        monkeypatch.setenv("KBN_DEPTH_FRONT_FUSION", "1")
        kb.ops.reload_env()
        assert not enc.pair_chain and dec.pair_chain and enc.fuse_s2d
        monkeypatch.setenv("KBN_NO_DEPTH_FRONT_FUSION", "1")
        kb.ops.reload_env()
        assert not enc.fuse_s2d, "the NO_ switch wins over the opt-in"
        enc.fuse_s2d = True
        assert enc.fuse_s2d, "an assignment overrides the environment"
        enc.fuse_s2d = None
        assert not enc.fuse_s2d
        assert kb.ops.knob("KBN_NO_SUCH_SWITCH") == 0
        # a switch set to something that is not a number ("true", "yes") is SET: it reads as 1 (the pre-kbn_knob host
        # switches treated any non-empty value but "0" as set; atoi alone would read 0 and silently leave the path on)
        for text, want in (("true", 1), ("yes", 1), ("0", 0), ("2", 2), (" 1", 1), ("", 0)):
            monkeypatch.setenv("KBN_NO_PAIR_TAIL", text)
            kb.ops.reload_env()
            assert kb.ops.knob("KBN_NO_PAIR_TAIL") == want, (text, want)
            assert dec.pair_tail == (want == 0)
    finally:
        monkeypatch.undo()
        kb.ops.reload_env()


def test_front_queries_answer_without_a_gpu():
    """The eligibility queries the host mirror asks before it launches anything of level 0 (kbn_kb1_front_query,
    kbn_kb1_depth_front_query, kbn_s2d_depth_front_query) are pure host logic."""
    assert kb.ops.kb1_front_supported(3, 48, 48, 352, 1216, 0.2)
    assert not kb.ops.kb1_front_supported(3, 32, 48, 352, 1216, 0.2), "only KBNet's 48 / 48 filters"
    assert not kb.ops.kb1_front_supported(3, 48, 48, 352, 1216, 1.5), "LeakyReLU as max(t, slope t) needs a slope in [0, 1]"
    assert kb.ops.kb1_front_supported(8, 16, 16, 352, 1216, 0.2, depth_branch=True)
    kitti, void = kb.kitti_config(), kb.void_config()
    assert kb.ops.s2d_depth_front_supported(2, kitti.min_pools, kitti.max_pools, 3, 8, 16, 16, 352, 1216, 0.2, 0.2)
    assert kb.ops.s2d_depth_front_supported(2, void.min_pools, void.max_pools, 3, 8, 16, 16, 480, 640, 0.2, 0.2)
    assert not kb.ops.s2d_depth_front_supported(2, [3, 5], [7], 3, 8, 16, 16, 64, 64, 0.2, 0.2), "only the compiled pool presets"
    assert not kb.ops.s2d_depth_front_supported(1, kitti.min_pools, kitti.max_pools, 3, 8, 16, 16, 352, 1216, 0.2, 0.2)
    os.environ["KBN_NO_SPLIT"] = "1"
    kb.ops.reload_env()
    try:
        assert not kb.ops.kb1_front_supported(3, 48, 48, 352, 1216, 0.2)
        assert not kb.ops.s2d_depth_front_supported(2, kitti.min_pools, kitti.max_pools, 3, 8, 16, 16, 352, 1216, 0.2, 0.2)
    finally:
        del os.environ["KBN_NO_SPLIT"]
        kb.ops.reload_env()


def test_latency_mode_host_logic():
    """KBNetModel.set_latency_mode (round 6) is host logic until a launch happens: the split-K rule is a function of the LAYER and of the
    mode's `frames` (never of a call's batch: a frame's bits inside a mode do not depend on how it is run), its ranges are never empty,
    the decoder blocks / KB levels that would launch split-K are the low-resolution ones, and switching the mode off leaves nothing behind."""
    import torch
    for cin in (64, 96, 128, 384, 768, 1024):
        for oc in (48, 64, 128, 256, 384):
            for hw in ((11, 38), (22, 76), (44, 152), (88, 304), (176, 608), (15, 20), (30, 40)):
                for stride, up in ((1, False), (2, False), (1, True)):
                    for frames in (1, 2, 4, 16):
                        ks = kb.ops.ksplit_for(cin, oc, hw[0], hw[1], stride, up2x=up, frames=frames)
                        chunks = cin // 16
                        assert 1 <= ks <= 16 and ks <= max(1, chunks // 2)
                        assert ks == 1 or -(-chunks // ks) * (ks - 1) < chunks, "no empty chunk range"
                        assert ks == kb.ops.ksplit_for(cin, oc, hw[0], hw[1], stride, up2x=up, frames=frames), "deterministic"
    assert kb.ops.ksplit_for(768, 256, 22, 76) > 1 and kb.ops.ksplit_for(768, 256, 22, 76, frames=16) == 1
    assert kb.ops.ksplit_for(128, 64, 176, 608) == 1, "a full-resolution layer fills the chip on its own"
    m = kb.modules.KBNetModel.from_config(kb.kitti_config(), torch.device("cpu"))
    d, e = m.decoder, m.encoder
    blocks = ((d.deconv4, (11, 38)), (d.deconv3, (22, 76)), (d.deconv2, (44, 152)), (d.deconv1, (88, 304)))
    levels = ((1, (176, 608)), (2, (88, 304)), (3, (44, 152)))
    assert not any(b.latency_splitk(*hw) for b, hw in blocks) and not any(e._image_splitk(l, *hw) for l, hw in levels), "off by default"
    m.set_latency_mode(True)
    assert [b.latency_splitk(*hw) for b, hw in blocks] == [True, True, True, False]
    assert [e._image_splitk(l, *hw) for l, hw in levels] == [False, True, True]
    m.set_latency_mode(True, frames=4)
    assert [b.latency_splitk(*hw) for b, hw in blocks] == [True, True, False, False]
    assert [e._image_splitk(l, *hw) for l, hw in levels] == [False, False, True]
    with pytest.raises(kb._lib.KbnError):
        m.set_latency_mode(True, frames=0)
    m.set_latency_mode(False)
    assert not any(b.latency_splitk(*hw) for b, hw in blocks) and m.latency_mode == 0
    assert all(getattr(c, "latency", 0) == 0 for top in m.modules() for c in top.modules())
