import sys, traceback, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import torch, quisk_amd as qh, pyoracle as oracle
import test_gpu_wdsp_shim_fuzz as T
try:
    T.test_random_walk_over_the_hand_off_with_the_caller_changing_sides(qh, oracle, int(sys.argv[1]))
    print("passed")
except BaseException:
    print(traceback.format_exc()[-1800:])
