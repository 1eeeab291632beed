"""A short run of the geometry fuzzer (tools/fuzz_gpu_geometry.py) under -m gpu: random sizes x sampling kinds x colour
transforms x coefficient classes through the batch kernels, one kind per batch and mixed, byte for byte against the oracle.
(The fuzzer found the one-MCU-wide 4:2:0 seam staging overflow in round 2; longer runs: tools/gpu_fuzz_geometry.sh.)"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _fuzzer(name="fuzz_gpu_geometry"):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", name + ".py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [2026, 2027])
def test_geometry_fuzz_bit_exact(seed):
    bad, total, paths = _fuzzer().run(seed, 80, verbose=False)
    assert total > 200 and bad == 0, (bad, total, paths)
    assert any(p.startswith("fused") for p in paths) and "mixed" in paths


def test_valid_stream_fuzz_bit_exact():
    """tools/fuzz_gpu_files.py: PIL-encoded streams of random size / sampling / quality / progressive / restart interval through
    Decoder (Worker route, also with scale()) and Pipeline (host and device entropy decoding) against the oracle's decode."""
    pytest.importorskip("PIL")
    assert _fuzzer("fuzz_gpu_files").run(77, 48, verbose=False) == 0


def test_extreme_geometries_bit_exact():
    """tools/extreme_sizes.py: 65535-wide, 65535-tall, one-pixel and 8191x4097 images of every fused kind through the batch path, at full
    size and (five kinds) at the reduced scales 4 / 2 / 1 of the band kernel
    (maximum sizes of SOF0: src/parser.rs:292-298)."""
    assert _fuzzer("extreme_sizes").run(verbose=False) == []


def test_worker_sequence_fuzz_bit_exact():
    """tools/fuzz_gpu_worker.py: random call sequences over the Worker boundary (prefixes of the rows, get_result / finish_plane,
    rows one by one / several at once, reduced IDCTs, the worker reused) — planes and pixels against the oracle."""
    assert _fuzzer("fuzz_gpu_worker").run(5, 120, verbose=False) == 0
