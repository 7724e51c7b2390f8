# A/B of the wave-priority hint in the third-level sweep loop (diagnostic library _tprio) and check of the fine-level default
for i in 1 2 3; do
  PATS_AMD_DIAG_LIB=1 python tools/bench_third.py 110136 2>/dev/null | grep "mode=kernel iters=100" | sed "s/^/third diag      /"
  PATS_AMD_DIAG_LIB=_tprio python tools/bench_third.py 110136 2>/dev/null | grep "mode=kernel iters=100" | sed "s/^/third diag+prio /"
done
python tools/fine_fused_time.py 20224
