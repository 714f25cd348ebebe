cd $GRAFT_REPO_ROOT
for cmd in "tools/fuzz_dfsdm.py 200 7" "tools/fuzz_parity.py 900 141" "tools/fuzz_strides.py 700 143" "tools/fuzz_receive.py 900 145" "tools/fuzz_live.py 600 155" "tools/fuzz_iq.py 500 147" "tools/fuzz_stream.py 240 149" "tools/fuzz_spectrum.py 500 151"; do
  echo "== python $cmd"
  timeout -k 10 1000 python $cmd 2>&1 | tail -1
done
g++ -std=c++17 -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/stubs/loopback_rccl.cpp -o /tmp/libloopback_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt 2>/dev/null
UC_TUNING=1 UC_RCCL_LIB=/tmp/libloopback_rccl.so UC_GROUP_SHARE_DEVICES=1 UC_LOOPBACK_FUZZ="400 157" timeout -k 10 900 python tests/group_loopback_child.py 2>&1 | tail -2
