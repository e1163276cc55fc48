# tools/gpu/rb_diag.sh <tag> "<flags1>" "<flags2>" ... : build + run variants of tools/probe/rb_diag.hip; timeline dumps under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
i=0
for f in "$@"; do
  i=$((i+1))
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 $f tools/probe/rb_diag.hip -o /tmp/rbd_$i 2>&1 | grep -E "error" ) &
done
wait
i=0
for f in "$@"; do
  i=$((i+1))
  echo "## variant $i: $f"
  timeout 120 /tmp/rbd_$i $O/timeline_$i.txt
done 2>&1 | tee $O/rb_diag.log
