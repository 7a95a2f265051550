"""Developer measurement: pinned-memory copy rates over PCIe, one direction at a time and both at once."""
import torch
n = 64 << 20
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, down, reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1):
                d_a.copy_(h_in, non_blocking=True)
        if down:
            with torch.cuda.stream(s2):
                h_out.copy_(d_b, non_blocking=True)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


run(True, True, 2)
print("H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s each way" % (run(True, False), run(False, True), run(True, True)))
