"""Evidence hygiene: the HBM-traffic figure bench.py quotes for the headline kernel must have been taken on the pixel
kernels' sources as they stand — profiles/round6/pmc_traffic.json records a sha256 over them (tools/kernel_sources.py) and the commit
of the tree the counter passes ran on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_sources  # noqa: E402


def test_headline_traffic_was_measured_on_the_current_pixel_kernels():
    f = os.path.join(ROOT, "profiles", "round6", "pmc_traffic.json")
    assert os.path.exists(f), "profiles/round6/pmc_traffic.json is missing: tools/gpu.sh traffic:1080p-420:fused420:s420_kernel"
    e = json.load(open(f))["1080p-420:fused420"]
    assert e.get("commit"), "the entry does not say which commit it was taken on"
    assert e["pixel_kernel_sources_sha256"] == kernel_sources.sha256(), (
        "the pixel kernels' sources changed after the counter passes of profiles/round6/pmc_traffic.json: take them again")
    assert 0.95 < e["hbm_bytes_per_decode"] / 3196846080 < 1.25  # (traffic against the algorithmic bytes of 256 x 1080p 4:2:0)


def test_bench_reports_the_provenance():
    sys.path.insert(0, ROOT)
    import bench
    t, src, prov = bench.measured_traffic("1080p-420", "fused420")
    assert t and "round6" in src and prov["traffic_commit"] and prov["traffic_taken_on_these_kernel_sources"] is True
