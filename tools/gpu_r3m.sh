#!/bin/bash
# round 3, call M: how the sync passes emit (JPGPU_EMIT_MODE builds: 2 = 16-byte stores from registers (default), 0 = 4-byte stores, (historical: those -D switches are gone, huff_sync_core.hpp keeps the table of results)
# 1 = no stores (cost of the bookkeeping; wrong output), 3 = 4-byte stores + LDS ring reader), and the expansion kernel with scalar control
O=gpurun_out/r3m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "pipeline or decoder or entropy or anchor" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 3 $O/pytest.log
run() { local name=$1; shift; env JPGPU_BATCH_KERNEL_TIMES=1 "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("256",):
    e = d["e2e"][k]
    print(sys.argv[2], k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("kernel_ms"), e["verified_vs_oracle"])
PY
}
run mode2 X=1
run mode0 JPGPU_LIBRARY=$PWD/jpeg-decoder_amd/libjpgpu_alt0.so
run mode1 JPGPU_LIBRARY=$PWD/jpeg-decoder_amd/libjpgpu_alt1.so
run mode3 JPGPU_LIBRARY=$PWD/jpeg-decoder_amd/libjpgpu_alt3.so
run mode2b X=1
run write JPGPU_SYNC_EMIT=0
