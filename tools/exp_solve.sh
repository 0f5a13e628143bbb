cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in 0; do
  BLUEST_DEBUG_SOLVE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp_m$mode -- python bench.py --no-cpu-baseline --steps 800 > gpurun_out/exp_m$mode.json 2> gpurun_out/exp_m$mode.err
  echo "mode $mode"; grep -h -E "k_solve_from_chunks|k_phi_chunks|k_grad_tiles" gpurun_out/exp_m$mode/*/*_kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)"/"/'
done
