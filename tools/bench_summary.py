#!/usr/bin/env python3
"""prints the figures of a bench.py JSON line that the round's work is steered by"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print('value %.3e  ms_per_step %.4f  frame median %.4f ms  setup %.1f s' % (d['value'], d['ms_per_step'], d['frame_ms_median'], d.get('setup_s', 0)))
print('K1 mean %.2f median %.2f min %.2f us over %d launches (region %s) frac %.3f' % (r['mean_launch_us'], r.get('median_launch_us') or 0, r.get('min_launch_us') or 0, r['launches_timed'], r.get('mean_launch_us_timed_region'), r['frac']))
print('kernel_us', d.get('kernel_us'), 'pipelined ms', d.get('pipelined', {}).get('ms_per_frame'))
if 'cpu_baseline' in d:
    print('cpu port %.3f ms  optimised %.3f ms  full-size %s' % (d['cpu_baseline']['ms_per_frame'], (d.get('cpu_optimised') or {}).get('ms_per_frame', 0), (d.get('full_size_check') or {}).get('ok')))
if 'far_8192' in d:
    f = d['far_8192']; print('far8192 sync %.4f ms kernels %s launch %s' % (f['frame_ms_median_sync'], f['kernel_us'], {k: v for k, v in f['launch_us'].items() if k != 'note'}))
    print('   scan frac %.3f pack frac %.3f frame frac %.3f' % (f['roofline']['scan']['frac'], f['roofline']['pack']['frac'], f['roofline']['frame']['frac']))
if 'configs_2' in d:
    c = d['configs_2']
    for k in ('far_1000', 'far_1000_tick_all', 'wide_camera', 'dense_tick_all'):
        t = c[k]['tick_roofline']
        print('%-18s sync %.4f ms ticked %d tick us sync %.2f async %.2f frac %.3f' % (k, c[k].get('frame_ms_median_sync') or 0, c[k]['entities_ticked'], t['us_sync_frames'], t['us_async_frames'], t['frac_async_frames']))
if 'lighting' in d:
    print('lighting %.1f us' % d['lighting']['kernel_us'])
