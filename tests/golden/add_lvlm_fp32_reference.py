"""Adds the EXACT-ARITHMETIC reference of config 3's front half to tests/golden/full_depth_cfg3_lvlm7b.pt (build container, CPU, ~3 GB of RAM):

    python tests/golden/add_lvlm_fp32_reference.py

A 28-layer random-weight decoder amplifies rounding noise: two correct bf16 implementations (torch CPU = the oracle, the HIP engine) that round at
the same points but sum in different orders end up several per cent apart after 28 layers, so "HIP vs the bf16 oracle" alone cannot tell a wrong
kernel from the noise floor.  The criterion the small-size tests already use (tests/test_qwen2_gpu.py) needs the same graph in fp32: HIP must be
no further from the exact result than 1.5 x what the bf16 oracle itself is.  This script streams the checkpoint layer by layer (the generator of
tests/full_depth_common.py yields tensors in parameter order, so only one layer is ever resident in fp32) through oracle/qwen2vl_ref.py's
decoder_layer and stores `prompt_hidden_fp32`, `output_hidden_fp32` and `aligner_out_fp32` next to the bf16 entries.
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from oracle import aligner_ref as A       # noqa: E402
from oracle import qwen2vl_ref as Q       # noqa: E402
import full_depth_common as C              # noqa: E402


def main():
    torch.set_num_threads(int(os.environ.get("TD_GOLDEN_THREADS", "8")))
    C.use_fast_host_generator()
    fn = os.path.join(HERE, "full_depth_cfg3_lvlm7b.pt")
    fx = torch.load(fn)
    qcfg = Q.Qwen2Config()
    rq = C.lvlm_request(qcfg.vocab, qcfg.hidden)
    ids = torch.tensor(rq["prompt_ids"] + rq["forced_ids"])
    n_p, n_o = len(rq["prompt_ids"]), len(rq["forced_ids"])
    nxt = int(rq["position_ids"].max()) + 1
    pos = torch.cat([rq["position_ids"], (nxt + torch.arange(n_o, dtype=torch.int32))[None].expand(3, n_o)], dim=1)
    gen = C.draw_qwen_weights(Q.param_shapes(qcfg))
    t0 = time.time()
    h, norm_w, layer, li = None, None, {}, 0
    cos = sin = None
    for name, t in gen:
        if name == "model.embed_tokens.weight":
            emb = torch.nn.functional.embedding(ids, t)
            emb[ids == C.IMAGE_PAD] = rq["vision_rows"]
            h = emb.float()                                   # the same bf16 inputs, exact arithmetic from here on
            cos, sin = Q.mrope_cos_sin(pos, qcfg, torch.float32)
        elif name == "model.norm.weight":
            norm_w = t.float()
        elif name.startswith("model.layers."):
            layer[name] = t.float()
            if len(layer) == 12:                              # q/k/v weight + bias, o, gate, up, down, two norms
                assert all(k.startswith(f"model.layers.{li}.") for k in layer)
                with torch.no_grad():
                    h, _, _ = Q.decoder_layer(layer, qcfg, li, h, cos, sin)
                print(f"layer {li}: {time.time() - t0:.0f} s", flush=True)
                layer, li = {}, li + 1
    assert li == qcfg.num_layers
    hid = Q.rms_norm(h, norm_w, qcfg.rms_eps)
    asd = {k: v.float() for k, v in C.draw_aligner_weights(A.param_shapes(qcfg.hidden, 4096)).items()}
    y = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(hid[n_p:], asd["mm_projector.0.weight"], asd["mm_projector.0.bias"])),
                                   asd["mm_projector.2.weight"], asd["mm_projector.2.bias"])
    pe = A.t5_layer_norm(y, asd["mm_projector.3.weight"])
    fx["prompt_hidden_fp32"], fx["output_hidden_fp32"], fx["aligner_out_fp32"] = hid[:n_p].contiguous(), hid[n_p:].contiguous(), pe.contiguous()

    def rel(a, b):
        return float((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt())
    print(f"bf16 oracle vs exact arithmetic: prompt hidden {rel(fx['prompt_hidden'], hid[:n_p]):.4f}, output hidden {rel(fx['output_hidden'], hid[n_p:]):.4f}, "
          f"aligner output {rel(fx['aligner_out'], pe):.4f}")
    torch.save(fx, fn)


if __name__ == "__main__":
    main()
