import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from colosseumrl_amd.batched import TTTBatch, BlokusBatch
dev = torch.device("cuda", 0)
def timed(fn, reps=10, rounds=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort(); return ts[len(ts) // 2]
print("TTT: dims k P B | roll8 roll512/step step_observe sample+step")
for dims, k, P in (((3,3),3,2), ((3,5),3,3), ((3,3,3),3,4), ((5,5),4,3), ((4,4),3,2), ((4,4),4,2), ((6,5),4,3), ((4,8),4,4), ((2,4,4),3,3), ((5,6),5,2)):
    for B in (262144, 262145, 1000):
        st = TTTBatch(dims, k, P, B, device=dev)
        r8 = timed(lambda: st.rollout(8, 1))
        r512 = timed(lambda: st.rollout(512, 1), reps=2, rounds=3) / 512
        so = timed(lambda: st.step_observe(None, 1, True))
        ss = timed(lambda: st.step(st.sample(1), auto_reset=True))
        print("%-10s k%d P%d %7d | %8.1f %8.4f %8.1f %8.1f" % (dims, k, P, B, r8, r512, so, ss), flush=True)
        del st
print("Blokus: B | roll8 roll64/step step_observe valid_list select is_valid")
for B in (16384, 16385, 16387, 1000, 4096, 65536):
    st = BlokusBatch(B, device=dev)
    st.rollout(24, 3)
    r8 = timed(lambda: st.rollout(8, 1), reps=3, rounds=3)
    r64 = timed(lambda: st.rollout(64, 1), reps=1, rounds=3) / 64
    so = timed(lambda: st.step_observe(None, 1, True), reps=5, rounds=3)
    out = torch.empty((B, 2048), dtype=torch.int32, device=dev)
    try:
        vl = timed(lambda: st.valid_list(2048, None, out), reps=5, rounds=3)
    except Exception as ex:
        vl = -1
    rank = torch.zeros((B,), dtype=torch.int32, device=dev)
    try:
        se = timed(lambda: st.select(rank), reps=5, rounds=3)
    except Exception as ex:
        se = -1
    print("%7d | %8.1f %8.2f %8.1f %8.1f %8.1f" % (B, r8, r64, so, vl, se), flush=True)
    del st, out
