"""CPU: the export module has the reference's state_dict keys, reproduces the golden vectors, and traces."""
import ast
import os

import numpy as np
import pytest
import torch

from mtgv import spec
from mtgv.export import EncoderModule, export_onnx, export_torchscript, to_torch_module

from conftest import GOLDEN


def _cfg(npz):
    d = ast.literal_eval(str(npz["cfg"]))
    return spec.EncoderConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()})


@pytest.mark.parametrize("name", ["micro_ae_conv_linear", "micro_ae_conv_act_mlp", "micro_ae_pool_mlp", "micro_plain"])
def test_keys_and_golden(name):
    g = np.load(os.path.join(GOLDEN, f"encoder_{name}.npz"))
    cfg = _cfg(g)
    m = EncoderModule(cfg)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s in spec.encoder_param_shapes(cfg).items()]
    m = to_torch_module(cfg, spec.random_encoder_state(cfg, 1))
    with torch.no_grad():
        z = m(torch.from_numpy(g["x"])).numpy()
    np.testing.assert_allclose(z, g["z_fp32"], atol=1e-5)


def test_torchscript_roundtrip(tmp_path):
    cfg = spec.EncoderConfig("ae", (96, 64), 3, 48, (1, 1, 2, 1), (8, 16, 32, 64), "conv+linear", True)
    sd = spec.random_encoder_state(cfg, 1)
    p = str(tmp_path / "enc.pt")
    export_torchscript(cfg, {f"model.encoder.{k}": v for k, v in sd.items()}, p)  # Lightning prefix accepted
    ts = torch.jit.load(p)
    x = torch.rand(3, 3, 96, 64)
    with torch.no_grad():
        np.testing.assert_allclose(ts(x).numpy(), to_torch_module(cfg, sd)(x).numpy(), atol=1e-6)
    try:
        import onnx  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError):
            export_onnx(cfg, sd, str(tmp_path / "enc.onnx"))


@pytest.mark.parametrize("arch", ["v8", "11"])
def test_detector_module_keys_and_oracle(arch, tmp_path):
    """detector export mirror (od_export.py:163-176): ultralytics key names, same numbers as the oracle, traces"""
    from mtgv.export_detector import DetectorModule, export_onnx as det_onnx, export_torchscript as det_ts, to_torch_module as det_module
    from oracle import detector_ref as D

    cfg = spec.yolo11_config(imgsz=64) if arch == "11" else spec.DetectorConfig(imgsz=64)
    want = spec.detector_param_shapes(cfg)
    have = [(k, tuple(v.shape)) for k, v in DetectorModule(cfg).state_dict().items() if not k.endswith("num_batches_tracked")]
    assert have == [(k, tuple(s)) for k, s in want.items()]
    sd = spec.random_detector_state(cfg, 3)
    frames = np.random.default_rng(4).integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    ref_pred, ref_protos = D.forward(sd, cfg, frames)
    x = D.preprocess(frames)
    m = det_module(cfg, sd)
    with torch.no_grad():
        pred, protos = m(x)
    np.testing.assert_allclose(pred.numpy(), ref_pred.numpy(), atol=2e-5)
    np.testing.assert_allclose(protos.numpy(), ref_protos.numpy(), atol=2e-5)
    p = str(tmp_path / "det.pt")
    det_ts(cfg, sd, p)
    ts = torch.jit.load(p)
    with torch.no_grad():
        tp, tq = ts(x)
    np.testing.assert_allclose(tp.numpy(), pred.numpy(), atol=1e-6)
    np.testing.assert_allclose(tq.numpy(), protos.numpy(), atol=1e-6)
    try:
        import onnx  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError):
            det_onnx(cfg, sd, str(tmp_path / "det.onnx"))
