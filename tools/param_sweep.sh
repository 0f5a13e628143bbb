for cfg in "20 5 8" "25 6 1" "16 4 3"; do
  for p in "ma_iterations=100" "ma_iterations=150" "ma_iterations=200" "ma_iterations=300" "ma_iterations=200 enter_per_round=40" "ma_iterations=200 support_init=2" "ma_iterations=150 enter_per_round=40"; do
    echo "$cfg | $p | $(python tools/colgen_run.py $cfg $p 2>&1 | tail -1 | cut -c1-150)"
  done
done
