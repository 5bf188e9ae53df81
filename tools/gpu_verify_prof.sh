cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_vf
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vf -o vf -- python $GRAFT_REPO_ROOT/tools/bench_verify.py Q4_K 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_vf.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_vf.log | cut -c1-300
cd $GRAFT_REPO_ROOT; python tools/prof_summary.py $(ls gpurun_out/prof_vf/*.db | head -1) 2>&1 | head -16
python - <<'PY'
import sqlite3, glob
con = sqlite3.connect(glob.glob("gpurun_out/prof_vf/*.db")[0]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
PY
