"""Forward / backward time of one summed bidirectional LSTM layer (csrc/bilstm.hip) at HAGCN's shapes: a single sequence of
batch x nodes steps (development aid).  usage: python tools/time_lstm.py [steps=3584]"""
import sys, time
import torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.hagcn import bilstm_sum

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
for I, H in [(1, 64), (64, 128), (128, 64)]:
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(I, H, 1, batch_first=True, bidirectional=True).to(dev)
    x = torch.rand(1, T, I, device=dev, requires_grad=True)
    w = torch.rand(1, T, H, device=dev)
    for _ in range(2):
        out = bilstm_sum(lstm, x); (out * w).sum().backward()
    torch.cuda.synchronize()
    n = 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        ev[0].record(); out = bilstm_sum(lstm, x); l = (out * w).sum(); ev[1].record(); l.backward(); ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    print(f"I {I:4d} H {H:4d} T {T}: forward {tf / n:7.3f} ms ({tf / n / T * 1e3:6.3f} us/step)  backward {tb / n:7.3f} ms ({tb / n / T * 1e3:6.3f} us/step)", flush=True)
