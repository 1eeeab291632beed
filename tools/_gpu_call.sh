cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tools/probe_d2h2.hip -o /tmp/probe_d2h2 && timeout 300 /tmp/probe_d2h2 16 > $O/probe_d2h2.txt 2>&1; cat $O/probe_d2h2.txt
