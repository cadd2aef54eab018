"""Binary index: walk time against where the allocator puts the copy -- python tools/probe_bot_placement.py [dummy MB allocated first]"""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import oracle
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dummy = torch.empty(mb << 20, dtype=torch.uint8, device="cuda") if mb else None
q = torch.from_numpy(oracle.synth_queries(1, 1024, 29523, 776, 1)).cuda()
def run(tag):
    idx = DeviceIndex.synthetic(0, 0, 21015324, 29523, 86, 1, 0, nat.VS_NONE)
    idx.search(q, 100); torch.cuda.synchronize()
    Profile.enable(True); Profile.reset()
    for _ in range(3): idx.search(q, 100)
    torch.cuda.synchronize()
    ms, n = Profile.read("csr_scan_topk"); Profile.enable(False)
    print(f"dummy {mb} MB, {tag}: walk {ms / n:.2f} ms", flush=True)
    idx.close()
run("first build")
run("second build")
