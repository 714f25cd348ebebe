cd $GRAFT_REPO_ROOT
run() { python3 bench.py "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4g %s  %.4f ms' % (d['value'], d['unit'], d.get('ms_per_step', 0)), flush=True)"
}
for rep in 1 2 3; do
for v in iq1024_bb iq1024 iq_bb; do
for ig in 4 8 16 32; do echo -n "$v group $ig: "; UC_TUNING=1 UC_IQ_GROUP=$ig run --variant $v; done
done
done
