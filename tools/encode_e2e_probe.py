"""End-to-end corpus-embed rate with tokenizer threads / processes (BERT-base shape, bf16, 1024 chunks of 512 tokens):
python tools/encode_e2e_probe.py [processes,...]   (a FILE: spawn-ed tokenizer workers re-import the main module)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from tools import bench_extras as bx
    dev = torch.device("cuda", 0)
    for tp in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,4").split(",")]:
        r, em = bx.encode_breakdown(torch, dev, "base", "bf16", 1024, tok_processes=tp, parity=False)
        print("tok_processes", tp, {k: r[k] for k in ("value", "tokenizer_only_chunks_per_s", "forward_only_chunks_per_s", "end_to_end_over_forward_only", "host_ms")}, flush=True)
        em.close()


if __name__ == "__main__":
    main()
