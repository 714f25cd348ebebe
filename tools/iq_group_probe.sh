set -e
for v in "UC_STATIC_DEAL=1" "UC_IQ_GROUP=64" "UC_IQ_GROUP=32" "UC_IQ_GROUP=16" "UC_IQ_GROUP=8"; do
  echo "== $v"
  env UC_TUNING=1 $v python bench.py --variant iq1024 --steps 20 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3e frames/s  %.3f ms/step  frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['frac']))"
done
