# effect of the pricing's entry tolerance (column enters when its reduced cost exceeds the level by this relative amount) on the certified gap and the time
for cfg in "12 12 1" "20 5 1" "20 5 8" "25 6 1" "16 4 3"; do
  for p in "enter_tol=1e-8" "enter_tol=1e-9" "enter_tol=1e-10" "enter_tol=1e-11" "enter_tol=0"; do
    echo "$cfg | $p | $(python tools/colgen_run.py $cfg $p 2>&1 | tail -1 | cut -c1-200)"
  done
done
