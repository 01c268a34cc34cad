"""Which cross-stream dependency shapes does hipGraph stream capture accept on this runtime?  (round 5: the single-graph form of the
streaming schedule segfaults inside hipStreamEndCapture.)  Each case runs in its own process: python tools/diag_capture_edges.py [case]"""
import subprocess
import sys

import torch

CASES = ["fork_join", "sibling_edge", "sibling_edge_advanced", "destroy_nonorigin_event", "chain_of_sibling_edges", "chain_keep_events", "nested_lanes_with_sibling_edge",
         "nested_lanes_keep_events", "many_nodes_keep_events", "stage_pipeline_forward_edges"]


def run(case):
    dev = torch.device("cuda")
    x = torch.ones(1 << 20, device=dev)
    A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    La, Lb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    y = [torch.zeros_like(x) for _ in range(8)]
    keep = []
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream(dev)
        A.wait_stream(cur)
        B.wait_stream(cur)
        if case == "fork_join":
            with torch.cuda.stream(A):
                y[0].copy_(x * 2)
            with torch.cuda.stream(B):
                y[1].copy_(x * 3)
        elif case in ("sibling_edge", "sibling_edge_advanced"):
            with torch.cuda.stream(A):
                y[0].copy_(x * 2)
                e = torch.cuda.Event()
                e.record(A)
                if case == "sibling_edge_advanced":
                    y[2].copy_(y[0] + 1)
            with torch.cuda.stream(B):
                B.wait_event(e)
                y[1].copy_(y[0] * 3)
        elif case == "stage_pipeline_forward_edges":
            # A -> B -> origin, several forward edges each, the origin forks lanes of its own: the shape of the stage-pipelined streaming graph
            ev = []
            for i in range(4):
                with torch.cuda.stream(A):
                    y[0].add_(x)
                    ea = torch.cuda.Event(); ea.record(A); ev.append(ea)
                with torch.cuda.stream(B):
                    B.wait_event(ea)
                    y[1].add_(y[0])
                    eb = torch.cuda.Event(); eb.record(B); ev.append(eb)
                cur.wait_event(eb)
                La.wait_stream(cur)
                with torch.cuda.stream(La):
                    y[2].add_(y[1])
                cur.wait_stream(La)
        elif case == "destroy_nonorigin_event":
            with torch.cuda.stream(A):
                y[0].copy_(x * 2)
                e = torch.cuda.Event()
                e.record(A)
            with torch.cuda.stream(B):
                B.wait_event(e)
                del e                                   # the event dies while the capture is still open
                y[1].copy_(y[0] * 3)
        elif case == "nested_lanes_keep_events":
            def ws(dst, src):                           # wait_stream with an event that outlives the capture
                e_ = torch.cuda.Event()
                e_.record(src)
                dst.wait_event(e_)
                keep.append(e_)
            with torch.cuda.stream(A):
                ws(La, A)
                with torch.cuda.stream(La):
                    y[0].copy_(x * 2)
                ws(A, La)
                y[2].copy_(y[0] + 1)
                e = torch.cuda.Event()
                e.record(A)
            with torch.cuda.stream(B):
                B.wait_event(e)
                ws(Lb, B)
                with torch.cuda.stream(Lb):
                    y[1].copy_(y[2] * 3)
                ws(B, Lb)
        elif case in ("chain_of_sibling_edges", "chain_keep_events", "many_nodes_keep_events"):
            n = 400 if case == "many_nodes_keep_events" else 6
            ea = eb = None
            for i in range(n):
                if case != "chain_of_sibling_edges":
                    keep.extend([ea, eb])
                with torch.cuda.stream(A):
                    if eb is not None:
                        A.wait_event(eb)
                    y[0].add_(y[1] if i else x)
                    ea = torch.cuda.Event()
                    ea.record(A)
                with torch.cuda.stream(B):
                    B.wait_event(ea)
                    y[1].add_(y[0])
                    eb = torch.cuda.Event()
                    eb.record(B)
        elif case == "nested_lanes_with_sibling_edge":
            with torch.cuda.stream(A):
                La.wait_stream(A)
                with torch.cuda.stream(La):
                    y[0].copy_(x * 2)
                A.wait_stream(La)
                y[2].copy_(y[0] + 1)
                e = torch.cuda.Event()
                e.record(A)
            with torch.cuda.stream(B):
                B.wait_event(e)
                Lb.wait_stream(B)
                with torch.cuda.stream(Lb):
                    y[1].copy_(y[2] * 3)
                B.wait_stream(Lb)
        cur.wait_stream(A)
        cur.wait_stream(B)
    g.replay()
    torch.cuda.synchronize()
    print(f"CAPTURE_EDGES {case}: ok ({[float(t[0]) for t in y[:3]]})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, c], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
            out = r.stdout.decode()
            line = [ln for ln in out.splitlines() if "CAPTURE_EDGES" in ln or "Fatal" in ln or "Error" in ln]
            print(f"CAPTURE_EDGES {c}: rc={r.returncode} {line[:2]}", flush=True)
