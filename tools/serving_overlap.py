#!/usr/bin/env python
"""How much the inference stream and the train stream really overlap on the device, from a rocprofv3 kernel trace of
tools/bench_serving.py: over the steady-state window, busy = union of all kernel intervals, sum = sum of their durations;
sum / busy > 1 means kernels of the two streams ran concurrently.  Also splits the time by side (train-step kernels vs
inference-step kernels, by kernel name).

  python tools/serving_overlap.py gpurun_out/prof/r03_serving_kernel_trace.csv [out.json]"""
import csv
import json
import sys

TRAIN = ('wsw_kernel', 'stackconv_wgrad', 'impala_loss', 'adam_flat', 'reduce_slices', 'ws_tab_kernel<4, 4, 1',
         'true, true, false', 'false, false, false, false, false, 2>', 'loss_finalize', 'global_norm')


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
  t0, t1 = iv[0][0], iv[-1][1]
  lo, hi = t0 + (t1 - t0) * 2 // 3, t1 - (t1 - t0) // 20            # the last third: steady state, past the captures
  iv = [x for x in iv if x[0] >= lo and x[1] <= hi]
  busy, cur_s, cur_e = 0, None, None
  for s, e, _ in iv:
    if cur_e is None or s > cur_e:
      if cur_e is not None:
        busy += cur_e - cur_s
      cur_s, cur_e = s, e
    else:
      cur_e = max(cur_e, e)
  busy += cur_e - cur_s
  total = sum(e - s for s, e, _ in iv)
  train = sum(e - s for s, e, n in iv if any(k in n for k in TRAIN))
  out = dict(window_ms=round((hi - lo) / 1e6, 2), kernels=len(iv), device_busy_frac=round(busy / (hi - lo), 3),
             sum_of_kernel_time_over_busy_time=round(total / busy, 3),
             kernel_time_per_wall_second=dict(total=round(total / (hi - lo), 3), backward_and_update_kernels=round(train / (hi - lo), 3)),
             note='sum / busy > 1: kernels of the inference stream and of the train stream overlap on the device; '
                  'backward_and_update_kernels counts kernels only the train step launches (forward kernels are shared '
                  'by both sides)')
  print(json.dumps(out))
  if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
  main()
