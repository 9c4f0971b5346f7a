"""Probe: MLP forward + autograd backward over T steps captured in a HIP graph (no simulator), replayed with new inputs."""
import torch, json, sys
dev, T = "cuda", 10
def nets(dt):
    mk = lambda *l: torch.nn.Sequential(*l).to(dev, dt)
    L, E = torch.nn.Linear, torch.nn.ELU
    return {"393-3": lambda: mk(L(393, 3)), "393-64-3": lambda: mk(L(393, 64), L(64, 3)), "393-64-elu-3": lambda: mk(L(393, 64), E(), L(64, 3)),
            "393-64-64-3 elu": lambda: mk(L(393, 64), E(), L(64, 64), E(), L(64, 3)), "16-64-3": lambda: mk(L(16, 64), L(64, 3)),
            "393-64-3 nobias": lambda: mk(L(393, 64, bias=False), L(64, 3, bias=False))}
for backend in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(backend)
    for dt in (torch.float64, torch.float32):
        for B in (1024, 2048):
            for name, make in nets(dt).items():
                din = 16 if name.startswith("16") else 393
                X = torch.randn(T, B, din, device=dev, dtype=dt)
                def run(net):
                    tot = X.new_zeros(())
                    for t in range(T): tot = tot + net(X[t]).pow(2).sum()
                    tot.backward(); return tot
                torch.manual_seed(0); net = make()
                side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        net.zero_grad(set_to_none=True); run(net)
                torch.cuda.current_stream().wait_stream(side); net.zero_grad(set_to_none=True)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side): run(net)
                worst = {}
                for k in range(4):
                    X.normal_(); g.replay(); torch.cuda.synchronize()
                    got = {n: p.grad.clone() for n, p in net.named_parameters()}
                    torch.manual_seed(0); ref = make(); run(ref)
                    for n, p in ref.named_parameters():
                        worst[n] = max(worst.get(n, 0.0), float((got[n] - p.grad).norm() / p.grad.norm()))
                bad = {n: v for n, v in worst.items() if v > 1e-4}
                print(json.dumps({"blas": backend, "dtype": str(dt)[6:], "B": B, "net": name, "bad_grads": bad}), flush=True)
