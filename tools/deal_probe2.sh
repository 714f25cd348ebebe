# static deal vs dynamic hand-out, per kernel family (bench.py side measurements); run on the GPU box
one() { env UC_TUNING=1 $1 python bench.py --variant $2 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-10s %-34s %.3e %s  %.3f ms/step  frac %.3f'%('$2','$1',d['value'],d['unit'],d['ms_per_step'],d['roofline']['frac']))"; }
for v in rx_real sync_cplx dechirp_down; do one UC_STATIC_DEAL=1 $v; one UC_STATIC_DEAL=0 $v; done
for v in iq1024 iq; do one UC_STATIC_DEAL=1 $v; one UC_IQ_GROUP=32 $v; one UC_IQ_GROUP=16 $v; done
one UC_STATIC_DEAL=1 stream; for c in 2 4 8 16; do one UC_STREAM_CHUNK=$c stream; done
