"""Input pipeline: the library's PNG reader, the reference-named loader functions
and the batched device loader against what the REFERENCE reads from the same files
(tests/golden/io/* written and read by tests/golden/gen_golden.py --only-io)."""
import glob
import os

import numpy as np
import pytest
import torch

import kbnet_amd as kb
from conftest import GOLDEN_DIR
from oracle import kbnet_oracle as orc

IO = os.path.join(GOLDEN_DIR, "io")
EXP = dict(np.load(os.path.join(GOLDEN_DIR, "io_expected.npz")))
IMAGES = [os.path.join(IO, f"triplet_{i}.png") for i in range(3)]
DEPTHS = [os.path.join(IO, f"depth_{i}.png") for i in range(3)]
KS = [os.path.join(IO, f"k_{i}.npy") for i in range(3)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("GPU tests need a visible MI355X (run with -m gpu on a GPU box)")
    kb._lib.load()
    return torch.device("cuda:0")


def read(path):
    with open(path, "rb") as f:
        return f.read()


# ------------------------------------------------------------------ host side (no GPU)
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(IO, "*.png"))))
def test_png_reader_matches_pil_bit_for_bit(path):
    from PIL import Image
    got = kb.loader.decode_png(read(path))
    im = Image.open(path)
    want = np.asarray(im.convert("RGB")) if im.mode == "P" else np.asarray(im)
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.array_equal(got, want)


def test_reference_named_loaders_match_the_reference():
    for name in ("gray8", "rgba", "palette"):
        p = os.path.join(IO, name + ".png")
        assert np.array_equal(kb.loader.load_image(p, normalize=True, data_format="HWC"), EXP[f"load_image_{name}_hwc"])
        assert np.array_equal(kb.loader.load_image(p, normalize=False, data_format="CHW"), EXP[f"load_image_{name}_chw_raw"])
    t1, t0, t2 = kb.loader.load_image_triplet(IMAGES[0], normalize=True)
    assert np.array_equal(t0, EXP["triplet_0_t"]) and np.array_equal(t1, EXP["triplet_0_tm1"]) and np.array_equal(t2, EXP["triplet_0_tp1"])
    assert np.array_equal(kb.loader.load_depth(DEPTHS[0]), EXP["load_depth_0_hw"])
    assert np.array_equal(kb.loader.load_depth(os.path.join(IO, "depth8.png"), data_format="CHW"), EXP["load_depth8_chw"])
    with pytest.raises(ValueError):
        kb.loader.load_depth(DEPTHS[0], data_format="NCHW")


def test_oracle_loader_matches_the_reference():
    for i in range(3):
        image, z, k = orc.load_inference_sample(IMAGES[i], DEPTHS[i], KS[i])
        assert np.array_equal(image, EXP[f"sample_{i}_image"])
        assert np.array_equal(z, EXP[f"sample_{i}_sparse_depth"])
        assert np.array_equal(k, EXP[f"sample_{i}_intrinsics"])
        assert image.dtype == z.dtype == k.dtype == np.float32


def test_batch_decode_matches_single_decode():
    files = [read(p) for p in IMAGES + DEPTHS]
    outs = [np.empty_like(kb.loader.decode_png(f)) for f in files]
    kb.loader.decode_png_batch(files, outs, threads=4)
    for f, o in zip(files, outs):
        assert np.array_equal(o, kb.loader.decode_png(f))
    with pytest.raises(kb._lib.KbnError):
        kb.loader.decode_png_batch([files[0][:100]], [outs[0]], threads=2)


def test_png_reader_error_behaviour():
    data = read(IMAGES[0])
    with pytest.raises(kb._lib.KbnError):
        kb.loader.decode_png(data[:200])                     # truncated
    with pytest.raises(kb._lib.KbnError):
        kb.loader.decode_png(b"not a png at all" * 4)
    bad = bytearray(data)
    bad[28] = 1                                              # IHDR interlace method -> Adam7: unsupported
    with pytest.raises(kb._lib.KbnError):
        kb.loader.png_info(bytes(bad))
    with pytest.raises(kb._lib.KbnError):
        kb.loader.decode_png(data, np.empty(10, np.uint8))   # short output buffer
    with pytest.raises(kb._lib.KbnError):
        kb.loader.InferenceFrameLoader(IMAGES, DEPTHS, KS, device=torch.device("cpu"))   # no CPU fallback


# ------------------------------------------------------------------------- device side
def test_png_reader_survives_mutated_files():
    """The reader parses files it did not write: 3000 mutations of the fixture PNGs (tests/png_fuzz.py) in a child process -- each call
    returns a status, none takes the process down."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "png_fuzz.py"), "3000"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "survived" in res.stdout, (res.returncode, res.stdout[-300:], res.stderr[-600:])


# ------------------------------------------------------------------ output side: data_utils.save_depth
SAVED = dict(np.load(os.path.join(GOLDEN_DIR, "saved_depth.npz")))


def test_save_depth_matches_the_reference_writer(tmp_path):
    """loader.save_depth against a file written by the REFERENCE's data_utils.save_depth (src/data_utils.py:154-167; gen_golden.py
    --only-save-depth) from the same depth map: same samples, and the reference's load_depth reads the same array from both -- incl.
    depths past the PNG's 16 bits (clipped at 65535, as PIL clips) and just below one sample (truncated, not rounded)."""
    z = SAVED["z"]
    path = str(tmp_path / "out.png")
    kb.loader.save_depth(z, path)
    mine, ref = kb.loader.decode_png(read(path)), kb.loader.decode_png(read(os.path.join(IO, "saved_depth.png")))
    assert mine.dtype == np.uint16 and np.array_equal(mine, ref)
    assert np.array_equal(mine, np.minimum(np.floor(z.astype(np.float64) * 256.0), 65535).astype(np.uint16))
    assert np.array_equal(kb.loader.load_depth(path), SAVED["loaded"])
    assert kb.loader.png_info(read(path)) == (z.shape[1], z.shape[0], 1, 16)
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(np.array(Image.open(path)), ref)   # a reader that is not ours agrees


@pytest.mark.parametrize("shape", [(1, 1), (3, 1), (2, 1300), (97, 33)])
@pytest.mark.parametrize("level", [0, 1, 9])
def test_png_encoder_round_trip_and_errors(shape, level):
    g = np.random.Generator(np.random.Philox(shape[1] + level))
    a = g.integers(0, 65536, size=shape, dtype=np.uint16)
    a.flat[0] = 65535
    data = kb.loader.encode_depth_png(a, level=level)
    assert np.array_equal(kb.loader.decode_png(data), a)
    lib = kb._lib.load()
    import ctypes as C
    cap = lib.kbn_png_encode_gray16_bound(shape[1], shape[0])
    buf, n = (C.c_ubyte * cap)(), C.c_size_t()
    args = (a.ctypes.data_as(C.c_void_p), shape[1], shape[0], buf)
    assert lib.kbn_png_encode_gray16(*args, cap - 1, C.byref(n), level) == kb._lib.KBN_ERR_WORKSPACE
    assert lib.kbn_png_encode_gray16(*args, cap, C.byref(n), 10) == kb._lib.KBN_ERR_INVALID_ARGUMENT
    assert lib.kbn_png_encode_gray16(None, shape[1], shape[0], buf, cap, C.byref(n), level) == kb._lib.KBN_ERR_INVALID_ARGUMENT
    assert lib.kbn_png_encode_gray16_bound(0, 5) == 0


@pytest.mark.gpu
def test_save_depth_batch_from_device(dev, tmp_path):
    """run_kbnet.py --save_outputs (reference src/kbnet.py:1018-1026) for a batch of device depth maps: one device pass
    (kbn_depth_to_u16_forward = np.uint32(z * 256) with PIL's clip), host threads write the files."""
    z = np.stack([SAVED["z"], SAVED["z"][::-1].copy(), SAVED["z"] * 0.5])
    zt = torch.from_numpy(z).unsqueeze(1).to(dev)
    assert np.array_equal(kb.loader.depth_samples(zt), kb.loader.depth_samples(z).reshape(zt.shape))
    bad = torch.tensor([float("nan"), -1.0, float("inf"), 255.998], device=dev)
    assert kb.loader.depth_samples(bad).tolist() == [0, 0, 65535, 65535]
    paths = [str(tmp_path / f"{i}.png") for i in range(3)]
    kb.loader.save_depth_batch(zt, paths, threads=3)
    assert np.array_equal(kb.loader.load_depth(paths[0]), SAVED["loaded"])
    for i in range(3):
        assert np.array_equal(kb.loader.decode_png(read(paths[i])), np.minimum(np.floor(z[i].astype(np.float64) * 256), 65535).astype(np.uint16))
    kb.loader.save_depth(zt[1], paths[0])   # a single device map
    assert np.array_equal(kb.loader.decode_png(read(paths[0])), kb.loader.decode_png(read(paths[1])))


@pytest.mark.gpu
def test_unpack_frames_bit_exact(dev):
    raw = np.stack([kb.loader.decode_png(read(p)) for p in IMAGES])            # 3 x H x 3W x 3 uint8
    dep = np.stack([kb.loader.decode_png(read(p)) for p in DEPTHS])            # 3 x H x W uint16
    w = raw.shape[2] // 3
    image, depth = kb.ops.unpack_frames(torch.from_numpy(raw).to(dev), torch.from_numpy(dep.view(np.int16)).to(dev),
                                        width=w, x_offset=w)
    for i in range(3):
        assert np.array_equal(image[i].cpu().numpy(), EXP[f"sample_{i}_image"])
        assert np.array_equal(depth[i].cpu().numpy(), EXP[f"sample_{i}_sparse_depth"])
    # RGBA input without a crop, 8-bit depth
    rgba = kb.loader.decode_png(read(os.path.join(IO, "rgba.png")))[None]
    image, _ = kb.ops.unpack_frames(torch.from_numpy(rgba).to(dev), None)
    assert np.array_equal(image[0].cpu().numpy(), EXP["single_image"])
    d8 = kb.loader.decode_png(read(os.path.join(IO, "depth8.png")))[None]
    _, depth = kb.ops.unpack_frames(None, torch.from_numpy(d8).to(dev))
    assert np.array_equal(depth[0].cpu().numpy(), EXP["load_depth8_chw"])


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 2, 3])
def test_frame_loader_batches_match_the_reference_dataset(dev, batch):
    loader = kb.loader.InferenceFrameLoader(IMAGES, DEPTHS, KS, use_image_triplet=True, batch_size=batch,
                                            device=dev, workers=3)
    assert len(loader) == -(-3 // batch)
    i = 0
    for image, sparse_depth, intrinsics in loader:
        assert image.is_cuda and image.shape[1:] == (3, 24, 40) and sparse_depth.shape[1:] == (1, 24, 40)
        for j in range(image.shape[0]):
            assert np.array_equal(image[j].cpu().numpy(), EXP[f"sample_{i}_image"])
            assert np.array_equal(sparse_depth[j].cpu().numpy(), EXP[f"sample_{i}_sparse_depth"])
            assert np.array_equal(intrinsics[j].cpu().numpy(), EXP[f"sample_{i}_intrinsics"])
            i += 1
    assert i == 3


@pytest.mark.gpu
def test_loader_feeds_the_forward(dev):
    """Files -> loader -> preprocess (row f1) -> forward: the reference's inference loop body
    (src/kbnet.py:887-921) on the device, against the oracle fed by the oracle's own loader."""
    cfg = kb.kitti_config().narrow()
    sds = kb.synthetic.make_state_dicts(cfg, seed=3)
    model = kb.modules.KBNetModel.from_config(cfg, dev)
    model.load_state_dicts(*sds)
    # the 24 x 40 fixtures are too small for five stride-2 levels: tile them to 96 x 160 on the device
    loader = kb.loader.InferenceFrameLoader(IMAGES, DEPTHS, KS, batch_size=3, device=dev)
    image, sparse, k = next(iter(loader))
    image, sparse = image.repeat(1, 1, 4, 4), sparse.repeat(1, 1, 4, 4)
    img, valid, filtered = kb.ops.preprocess(image, sparse)
    out = model.forward(img, filtered, valid, k)
    samples = [orc.load_inference_sample(IMAGES[i], DEPTHS[i], KS[i]) for i in range(3)]
    oimg = torch.from_numpy(np.stack([s[0] for s in samples])).repeat(1, 1, 4, 4)
    osp = torch.from_numpy(np.stack([s[1] for s in samples])).repeat(1, 1, 4, 4)
    ok = torch.from_numpy(np.stack([s[2] for s in samples]))
    ofilt, ovalid = orc.validity_and_outlier_removal(osp)
    ref = orc.kbnet_forward(oimg / 255.0, ofilt, ovalid, ok, *sds, cfg.min_pools, cfg.max_pools,
                            cfg.min_predict_depth, cfg.max_predict_depth)
    assert float((out.cpu() - ref).abs().max() / ref.abs().max()) < 1e-4
