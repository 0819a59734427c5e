cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for mt in 1 2 3 8; do
  SEFD_LSTM_MT=$mt timeout 300 python bench.py --model fullsubnet --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/r2_run13_mt$mt.log 2>&1
  echo "MT=$mt $(tail -1 $O/r2_run13_mt$mt.log | cut -c1-140)"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fsn2 -o fsn -- python $GRAFT_REPO_ROOT/bench.py --model fullsubnet --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_run13_prof.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/prof_fsn2'
f=glob.glob(O+'/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ls=[(r['Kernel_Name'][:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Grid_Size_X'] if 'Grid_Size_X' in r else '') for r in rows if 'lstm' in r['Kernel_Name'] and 'mark' not in r['Kernel_Name']]
for x in ls[-16:]: print(x)
PY
find $O/prof_fsn2 -name "*kernel_trace*" -delete
head -12 $O/prof_fsn2/fsn_kernel_stats.csv | cut -c1-160
