"""GPU half of tests/test_reference_seam.py.  The reference tree does not travel to the GPU box, so the reference's own classes are driven on the CPU box
over this build's `exllamav3_ext` module with a recording op layer (tests/golden/make_seam_fixture.py) and the recorded call sequences are replayed here
on cuda:0 and compared with the oracle.  (Where a GPU and /root/reference meet, `python tests/_reference_seam_script.py cuda:0` with
PYTHONPATH = stub dir : repo : reference runs the classes live.)"""
import glob
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
STUB_DIR = os.path.join(ROOT, "exllamav3_amd", "stub")


FIXTURE = os.path.join(ROOT, "tests", "golden", "ref_seam_calls.npz")


def test_recorded_reference_call_sequences_replay_on_the_gpu(dev):
    """Runs on every GPU box (no reference tree needed): tests/golden/ref_seam_calls.npz holds, for the reference's own LinearEXL3.forward (kernel
    route at 1 / 16 rows, reconstruct + hgemm at 145, fused reconstruct at 1030; mul1 and 3INST), RMSNorm.forward and RoPE.apply, the exact
    sequence of `exllamav3_ext` ops those classes issued over this build's module -- recorded on the CPU box by tests/golden/make_seam_fixture.py,
    arguments with their storage aliasing -- plus the ORACLE's result for the same inputs.  The sequences are replayed here through
    exllamav3_amd.ext on cuda:0 and the tensor the reference method returned is compared with the oracle."""
    import json
    import numpy as np
    import torch
    from exllamav3_amd import ext
    z = np.load(FIXTURE)
    cases = json.loads(bytes(z["meta"]).decode())
    DT = {"f2": torch.float16, "f4": torch.float32, "i2": torch.int16, "i4": torch.int32, "i8": torch.int64}
    assert len(cases) >= 10
    ops = set()
    for name, c in cases.items():
        stor = {}
        for s in c["storages"]:
            buf = torch.zeros((s["nbytes"] + 15) // 16 * 16, dtype=torch.uint8, device=dev)
            if s["has_init"]:
                init = torch.from_numpy(z[f"{name}/s{s['id']}"].copy()).to(dev)
                buf[: init.numel()].copy_(init)
            stor[s["id"]] = buf

        def tensor(d):
            dt = DT[d["dt"]]
            base = stor[d["s"]].view(dt)
            return torch.as_strided(base, d["shape"], d["strides"], d["off"])

        def arg(a):
            return tensor(a["t"]) if "t" in a else a["v"]

        for call in c["calls"]:
            ops.add(call["op"])
            getattr(ext, call["op"])(*[arg(a) for a in call["args"]], **{k: arg(v) for k, v in call["kwargs"].items()})
        torch.cuda.synchronize()
        got = tensor(c["result"]).float().cpu().numpy()
        ref = z[f"{name}/expect"].astype(np.float32).reshape(got.shape)
        assert np.isfinite(got).all(), name
        err = float(np.abs(got - ref).max() / (np.sqrt((ref ** 2).mean()) + 1e-9))
        assert err < c["tol"], (name, err, c["tol"])
    assert {"exl3_gemm", "had_r_128", "reconstruct", "reconstruct_had_slice", "hgemm", "rms_norm", "rope"} <= ops
