"""Which hardware queue do torch's streams land on, and do two streams overlap?  (rocprofv3 --kernel-trace of this script:
tools/trace_timeline.py shows Queue_Id and the start / end times.)  GPU box."""
import sys, torch
x = torch.zeros(1 << 26, device="cuda")          # 256 MB: a fill takes ~100 us
ys = [torch.zeros(1 << 26, device="cuda") for _ in range(3)]
streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(priority=-1)]
torch.cuda.synchronize()
for rep in range(3):
    for k, s in enumerate(streams):
        with torch.cuda.stream(s):
            (x if k == 0 else ys[k - 1]).fill_(float(rep))
    torch.cuda.synchronize()
print("done")
