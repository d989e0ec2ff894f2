cd $GRAFT_REPO_ROOT
for q in 4 8 4 8; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --no-prove --no-cpu 2>/dev/null > gpurun_out/hwq_$q.json
python -c "
import json; j=json.load(open('gpurun_out/hwq_$q.json')); print('queues $q', round(j['value'],4), round(j['ms_per_step'],4), j['roofline']['avg_launch_ms'])"
done
