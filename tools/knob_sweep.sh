cd $GRAFT_REPO_ROOT
run() { # env... -- args
  python3 bench.py "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.4g %s  %.4f ms' % (d['value'], d['unit'], d.get('ms_per_step', 0)), flush=True)"
}
for rep in 1 2; do
for ch in 1 2 4; do echo -n "stream chunk $ch: "; UC_TUNING=1 UC_STREAM_CHUNK=$ch run --variant stream; done
for g in 512 768 1024; do echo -n "stream grid $g: "; UC_TUNING=1 UC_GRID=$g run --variant stream; done
for bg in 16 32 64; do echo -n "band group $bg: "; UC_TUNING=1 UC_BAND_GROUP=$bg run --no-configs --no-hello1 --no-receive --no-cpu-baseline --no-live-traffic; done
for g in 1280 1536 2048; do echo -n "band grid $g: "; UC_TUNING=1 UC_GRID=$g run --no-configs --no-hello1 --no-receive --no-cpu-baseline --no-live-traffic; done
for ig in 16 32 64; do echo -n "iq1024_bb group $ig: "; UC_TUNING=1 UC_IQ_GROUP=$ig run --variant iq1024_bb; done
done
