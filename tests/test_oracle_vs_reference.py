"""Build container only: oracle == imported reference at the full KITTI / VOID sizes
(bitwise), including the intrinsics-scaling quirk.  Skipped where /root/reference is absent
(e.g. on the GPU box), where the committed golden vectors take over."""
import os
import sys
import types

import pytest
import torch

from oracle import kbnet_oracle as orc
import kbnet_amd as kb

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def _reference_model(cfg):
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import gen_golden
    return gen_golden.build_reference_model(cfg), gen_golden.load


@pytest.mark.parametrize("preset,shape,options", [("kitti", (352, 1216), {}), ("void", (480, 640), {}),
                                                  # run_kbnet.py --deconv_type transpose / --activation_func elu | sigmoid
                                                  ("kitti", (352, 1216), {"deconv_type": "transpose", "activation_func": "elu"}),
                                                  ("void", (480, 640), {"activation_func": "sigmoid"})])
def test_full_size_bitwise(preset, shape, options):
    import dataclasses
    cfg = dataclasses.replace(kb.PRESETS[preset](), **options)
    model, load = _reference_model(cfg)
    sds = kb.synthetic.make_state_dicts(cfg, seed=0, gain=kb.synthetic.PARITY_GAIN[preset] if options else 1.0)
    load(model.sparse_to_dense_pool, sds[0])
    load(model.encoder, sds[1])
    load(model.decoder, sds[2])
    model.eval()
    image, sparse, valid, k = kb.synthetic.make_frames(1, *shape, preset, seed=1)
    with torch.no_grad():
        ref = model.forward(image, sparse, valid, k)
    out = orc.kbnet_forward(image, sparse, valid, k, *sds, cfg.min_pools, cfg.max_pools,
                            cfg.min_predict_depth, cfg.max_predict_depth, slope=orc.activation_slope(cfg.activation_func))
    assert torch.equal(ref, out)
    assert float(out.std()) > 0
