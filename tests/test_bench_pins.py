"""tests/golden/bench_pins.json -- the oracle logits bench.py's correctness gate compares the timed pipeline with -- re-derived here from the oracle
(CPU, no GPU): the committed file cannot drift from oracle/exl3_oracle.py or from SyntheticEXL3Llama.pin_model's host-seeded tensors."""
import json
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def _pins():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_pins.json")))["pins"]


def test_every_driver_line_config_has_a_pin():
    import make_bench_pins as mk
    from exllamav3_amd.llama_path import SyntheticEXL3Llama
    pins = _pins()
    assert sum(1 for c in mk.CONFIGS if len(c) > 4) >= 3          # the with-attention lines are pinned too (round 6)
    for cfg in mk.CONFIGS:
        name, K, cb, bsz = cfg[:4]
        key = SyntheticEXL3Llama.pin_key(name, K, cb, bsz, cfg[4] if len(cfg) > 4 else None)
        assert key in pins and pins[key]["shape"] == [bsz, SyntheticEXL3Llama.PIN_VOCAB] and len(pins[key]["logits"]) == bsz * SyntheticEXL3Llama.PIN_VOCAB
        assert 0.1 < pins[key]["rms"] < 10.0 and np.isfinite(pins[key]["logits"]).all()


@pytest.mark.parametrize("idx", range(8))
def test_pins_equal_the_oracle(idx):
    """Each pin == the oracle composition over the pin model, value for value (same numpy code on the same host-seeded tensors: exact up to the
    float32 -> JSON -> float32 round trip, which is lossless)."""
    import make_bench_pins as mk
    cfg = mk.CONFIGS[idx]
    r = mk.compute(cfg)
    pin = _pins()[r["key"]]
    got, want = np.asarray(r["logits"], dtype=np.float32), np.asarray(pin["logits"], dtype=np.float32)
    assert r["shape"] == pin["shape"]
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, float(np.abs(want).max()))


def test_pin_model_is_seed_deterministic_and_host_generated():
    """Two builds of the pin model are bit-identical (numpy PCG64 stream, no device generator involved) and differ between codebooks / bitrates."""
    import torch
    from exllamav3_amd.llama_path import SyntheticEXL3Llama
    a = SyntheticEXL3Llama.pin_model("llama-3.2-1b", 4, 2, "cpu", 1)
    b = SyntheticEXL3Llama.pin_model("llama-3.2-1b", 4, 2, "cpu", 1)
    c = SyntheticEXL3Llama.pin_model("llama-3.2-1b", 4, 0, "cpu", 1)
    for nm in ("q", "down"):
        assert torch.equal(a.layers[0][nm].trellis, b.layers[0][nm].trellis) and torch.equal(a.layers[0][nm].svh, b.layers[0][nm].svh)
        assert not torch.equal(a.layers[0][nm].trellis, c.layers[0][nm].trellis)
    assert torch.equal(a.x0, b.x0) and a.n_layers == 1 and a.shape.vocab == SyntheticEXL3Llama.PIN_VOCAB and int(a.positions[0]) == SyntheticEXL3Llama.PIN_POS
