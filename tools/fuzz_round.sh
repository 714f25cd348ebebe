cd $GRAFT_REPO_ROOT
echo "Round 4 (commit $(git rev-parse --short HEAD 2>/dev/null)); every tool runs the suite's bars against the oracle on random draws (new seeds):"
for cmd in "tools/fuzz_parity.py 600 41" "tools/fuzz_strides.py 400 43" "tools/fuzz_receive.py 800 45" "tools/fuzz_iq.py 300 47" "tools/fuzz_stream.py 160 49" "tools/fuzz_spectrum.py 300 51"; do
  echo "== python $cmd"
  timeout -k 10 900 python $cmd 2>&1 | tail -1
done
echo "== python tools/soak.py"
timeout -k 10 600 python tools/soak.py 2>&1 | tail -12
