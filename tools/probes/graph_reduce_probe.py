"""Probe (round 6): is a captured multi-block reduction replayed correctly?  `x.sum(1)` of [n, L, 768] fp32 (the mean pooling of
SentenceEncoder.forward) captured in a graph, replayed three times with fresh inputs, against the eager sum."""
import torch


def main():
    dev = torch.device("cuda:0")
    for n, L in ((64, 512), (64, 504), (16, 512), (16, 1024), (128, 256), (4, 4096)):
        x = torch.randn(n, L, 768, device=dev)
        m = (torch.rand(n, L, 1, device=dev) > 0.3).float()
        variants = {
            "(x * m).sum(1)": lambda: (x * m).sum(1),
            "x.sum(1)": lambda: x.sum(1),
            "x * m (no reduction)": lambda: x * m,
            "(x * m).sum(1) / m.sum(1)": lambda: (x * m).sum(1) / m.sum(1).clamp(min=1e-9),
            "m.sum(1)": lambda: m.sum(1),
            "bmm": lambda: torch.bmm(m.transpose(1, 2), x).squeeze(1),
        }
        if L % 64 == 0:
            variants["two-stage sum"] = lambda: (x * m).view(n, -1, 64, 768).sum(2).sum(1)
        for name, fn in variants.items():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = fn()
            res = []
            for r in range(3):
                x.normal_()
                g.replay()
                torch.cuda.synchronize()
                ref = fn()
                res.append(float((out - ref).abs().max()))
            print(f"n={n} L={L} {name}: replay - eager (fresh input each time) {res}", flush=True)


if __name__ == "__main__":
    main()
