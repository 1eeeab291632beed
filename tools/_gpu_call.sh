cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "host_light" > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log
bash tools/gpu.sh r5j bench
