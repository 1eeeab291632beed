cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "host_light or progressive" > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
for m in 0 1; do echo "== JPGPU_PIPE_HOST_LIGHT=$m"; for n in 4096 256; do JPGPU_PIPE_HOST_LIGHT=$m python tools/pipe_calls.py --images $n --calls 8 2>&1 | tail -1 | cut -c 1-120; done; done > $O/light_ab.txt 2>&1
cat $O/light_ab.txt
