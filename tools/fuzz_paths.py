"""Randomised sweep, fourth part: the PATHS against one another at random sizes -- the inference executor against the
module-by-module path and the training executor against the op tape (both bit for bit: outputs, clusters, every gradient),
through the bodies of tests/test_gpu_model.py with random mesh frequencies, pooling types and the depth head; the pooling
layer with 1 / 2 / 3 matching steps and every edge-weight type against the oracle.   python tools/fuzz_paths.py [seconds] [seed]"""
import os, sys, time, random, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_model as M

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda:0')
count, fails = {}, []
t_end = time.time() + budget


def run(tag, fn, *args):
    count[tag] = count.get(tag, 0) + 1
    try:
        fn(dev, *args)
    except Exception:                                        # noqa: BLE001 -- a sweep reports and goes on
        fails.append((tag, args, traceback.format_exc().strip().splitlines()[-1][:200]))


while time.time() < t_end:
    n = rng.choice([2, 3, 4, 5, 7, 9, 10, 13, 17, 21, 26])
    pool = rng.choice(['max', 'max', 'mean'])
    depth = rng.random() < 0.25
    run('inference executor == module path', M.test_whole_network_executor_equals_module_path, n, pool, depth)
    run('training executor == op tape', M.test_training_executor_equals_op_tape, min(n, 17), pool, depth)
    run('pooling layer, other step counts', M.test_pooling_layer_other_step_counts_against_oracle, rng.choice([1, 2, 3]), pool)
    run('edge weight types', M.test_edge_weight_types_match_oracle, rng.choice([-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]))
print('cases', count, 'failures', len(fails))
for f in fails[:12]:
    print('  FAIL', f)
sys.exit(1 if fails else 0)
