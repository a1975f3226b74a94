export PYTHONPATH=/root/repo; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
PALU_PVQ_DIRECT=0 timeout 600 python -m pytest /root/repo/tests/test_quant_decode_gpu.py -x -q -m gpu -k "24_code" 2>&1 | tail -15
for cfg in "3 384 65536" "4 192 131072"; do
rocprofv3 --kernel-trace --stats -d /tmp/prof_pvq -o pvq -- python /root/repo/tools/time_pvq_loop.py $cfg 20 > /tmp/prof_pvq.log 2>&1
tail -1 /tmp/prof_pvq.log
find /tmp/prof_pvq -name "*.csv" | head
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_pvq/**/*kernel_stats.csv', recursive=True)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'pv_' in r['Name']:
            print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
rm -rf /tmp/prof_pvq
done
