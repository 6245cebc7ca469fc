"""More seeds of every seeded setter walk of the suite in one process: the (qh, oracle, seed) walks as they are, the Quisk api / bank / rx walks
with (mode, fs, play) drawn from the combinations their tests list.  usage: walk_sweep_all.py <first> <last> [family ...]
(families: rxa rxa_replay rxa_long rxa_notch names shim ana ana_bank dropin api api_wdsp bank rx; default all; env CHARS: how much of a
failure's message to print; QH_COMBOS=random: the Quisk api / bank walks with (mode, fs, play) drawn per seed)"""
import importlib, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch          # before libquiskhip: one HIP runtime per process
import quisk_amd as qh
import pyoracle as oracle

def mod(n): return importlib.import_module(n)
COMBOS = [(3, 192000, 48000), (3, 111111, 96000), (4, 96000, 48000), (5, 192000, 48000), (3, 48000, 48000), (1, 133333, 48000), (4, 185185, 96000),
          (5, 96000, 192000), (3, 192000, 192000), (1, 48000, 96000), (3, 370370, 48000), (5, 53333, 48000), (0, 96000, 48000), (2, 192000, 48000),
          (7, 192000, 96000), (8, 111111, 48000), (9, 192000, 48000), (13, 96000, 48000), (10, 48000, 48000)]
if os.environ.get("QH_COMBOS") == "random":         # (mode, fs, play) drawn per seed from every rate family the planner knows, not the tests' own list
    import numpy as _np
    _MODES = [int(v) for v in os.environ.get("QH_COMBOS_MODES", "0,1,2,3,4,5,7,8,9,10,13").split(",")]
    _FS = [48000, 53333, 96000, 111111, 133333, 185185, 192000, 240000, 250000, 370370, 384000, 480000, 740740, 960000]
    class _Combos:
        def __len__(self): return 1 << 30
        def __getitem__(self, i):
            r = _np.random.default_rng(1234567 + i)
            return (int(r.choice(_MODES)), int(r.choice(_FS)), int(r.choice([48000, 96000, 192000])))
    COMBOS = _Combos()
def rx_combos():
    from quisk_amd import rxfilter as r
    return [(96000, r.USB), (192000, r.LSB), (48000, r.CWU), (96000, r.AM), (96000, r.FM), (240000, r.USB), (192000, r.AM), (48000, r.USB),
            (192000, r.CWL), (48000, r.AM), (192000, r.FM), (250000, r.LSB)]
def rx_random(s):
    from quisk_amd import rxfilter as r
    g = _np.random.default_rng(7654321 + s)
    return (int(g.choice(_FS)), [r.USB, r.LSB, r.CWU, r.CWL, r.AM, r.FM][int(g.integers(0, 6))])
FAM = {
    "rxa": lambda s: mod("test_gpu_rxa_fuzz").test_random_setter_walk(qh, oracle, s),
    "rxa_replay": lambda s: mod("test_gpu_rxa_fuzz").test_random_setter_walk_block_at_a_time_with_graph_replay(qh, oracle, s),
    "rxa_long": lambda s: mod("test_gpu_rxa_fuzz").test_random_setter_walk_with_long_filters_agc_windows_and_long_calls(qh, oracle, s),
    "rxa_notch": lambda s: mod("test_gpu_rxa_fuzz").test_random_setter_walk_with_the_notch_database_the_lms_sizes_and_fm(qh, oracle, s),
    "names": lambda s: mod("test_gpu_wdsp_names_fuzz").test_random_walk_through_the_wdsp_names(qh, oracle, s),
    "shim": lambda s: mod("test_gpu_wdsp_shim_fuzz").test_random_walk_over_the_hand_off_with_the_caller_changing_sides(qh, oracle, s),
    "ana": lambda s: mod("test_gpu_analyzer_fuzz").test_random_walk_over_the_display_engine(qh, oracle, s),
    "ana_bank": lambda s: mod("test_gpu_analyzer_fuzz").test_random_walk_over_a_bank_of_displays(qh, oracle, s),
    "dropin": lambda s: mod("test_gpu_filter_dropin_fuzz").test_one_struct_between_this_library_and_the_reference(qh, oracle, s),
    "api": lambda s: mod("test_gpu_quisk_api_fuzz").test_random_setter_walk_over_the_one_receiver_api(qh, oracle, s, *COMBOS[s % len(COMBOS)]),
    "api_wdsp": lambda s: mod("test_gpu_quisk_api_fuzz").test_random_setter_walk_with_wdsp_in_the_audio_path(qh, oracle, s, *COMBOS[s % len(COMBOS)]),
    "bank": lambda s: mod("test_gpu_quisk_bank_fuzz").test_random_setter_walk_over_the_bank(qh, oracle, s, *COMBOS[s % len(COMBOS)]),
    "rx": lambda s: mod("test_gpu_quisk_rx_fuzz").test_random_walk(qh, oracle, *(rx_random(s) if os.environ.get("QH_COMBOS") == "random" else rx_combos()[s % 12]), s),
}
a, b = int(sys.argv[1]), int(sys.argv[2])
fams = sys.argv[3:] or list(FAM)
for fam in fams:
    bad, t0 = 0, time.time()
    for seed in range(a, b + 1):
        if os.environ.get("QH_PROGRESS"): print("seed", seed, flush=True)
        try:
            FAM[fam](seed)
        except AssertionError as e:
            bad += 1
            print("%s seed %d: %s" % (fam, seed, str(e)[:int(os.environ.get("CHARS", "300"))].replace("\n", " ")), flush=True)
        except BaseException as e:
            if isinstance(e, KeyboardInterrupt): raise
            bad += 1
            print("%s seed %d: %s" % (fam, seed, traceback.format_exc()[-600:].replace("\n", " | ")), flush=True)
    print("== %s: %d walks, %d bad, %.0f s" % (fam, b - a + 1, bad, time.time() - t0), flush=True)
