cd $GRAFT_REPO_ROOT
echo "Round ${UC_ROUND:-5} (commit $(git rev-parse --short HEAD 2>/dev/null)); every tool runs the suite's bars against the oracle on random draws (new seeds):"
for cmd in "tools/fuzz_parity.py 600 ${UC_SEED:-41}" "tools/fuzz_strides.py 400 $((${UC_SEED:-41}+2))" "tools/fuzz_receive.py 800 $((${UC_SEED:-41}+4))" "tools/fuzz_live.py 400 $((${UC_SEED:-41}+14))" "tools/fuzz_iq.py 300 $((${UC_SEED:-41}+6))" "tools/fuzz_stream.py 160 $((${UC_SEED:-41}+8))" "tools/fuzz_spectrum.py 300 $((${UC_SEED:-41}+10))" "tools/fuzz_dfsdm.py 200 $((${UC_SEED:-41}+12))"; do
  echo "== python $cmd"
  timeout -k 10 900 python $cmd 2>&1 | tail -1
done
echo "== group logic at random world sizes (loop-back stand-in for RCCL): tests/group_loopback_child.py, UC_LOOPBACK_FUZZ"
g++ -std=c++17 -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stubs/loopback_rccl.cpp -o /tmp/libloopback_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt 2>/dev/null
UC_TUNING=1 UC_RCCL_LIB=/tmp/libloopback_rccl.so UC_GROUP_SHARE_DEVICES=1 UC_LOOPBACK_FUZZ="150 53" timeout -k 10 900 python tests/group_loopback_child.py 2>&1 | tail -2
echo "== python tools/soak.py"
timeout -k 10 600 python tools/soak.py 2>&1 | tail -12
