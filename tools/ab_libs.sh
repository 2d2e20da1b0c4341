#!/bin/bash
# Same-box A/B of library builds: tools/ab_libs.sh <reps> <lib name> ...
# (tools/probe/_lib/lib_<name>.so), alternating runs of the default bench.
reps=$1; shift
for rep in $(seq $reps); do
  for l in "$@"; do
    TIKE_AMD_LIB=$PWD/tools/probe/_lib/lib_$l.so python3 bench.py --no-cpu-baseline --no-secondary --breakdown --steps ${STEPS:-10} ${WL:+--workload $WL} 2>&1 |
      python3 -c "
import sys, json, re
k = {}
for line in sys.stdin:
    m = re.match(r'\s+(tike_\w+)\s+calls\s+\d+ avg\s+([\d.]+) ms', line)
    if m: k[m.group(1)] = float(m.group(2))
    if line.startswith('{'): v = json.loads(line)['value']
names = ['tike_poisson_steps_grad_ifft2_pass1', 'tike_fwd_grad_ifft2_pass1_slices', 'tike_probe_preconditioner', 'tike_position_sums', 'tike_fwd_pass1', 'tike_fwd_gradient_scale', 'tike_grad_ifft2_pass1', 'tike_fwd_grad_ifft2_pass1', 'tike_ifft2_pass2_gradients', 'tike_lstsq_step_stats', 'tike_scatter_patches', 'tike_psi_preconditioner', 'tike_eigen_position_sums1', 'tike_eigen_pixel_update1']
print('%-10s %8.0f  ' % ('$l', v) + '  '.join('%.3f' % k.get(n, 0) for n in names))
"
  done
done
