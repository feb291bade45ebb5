#!/usr/bin/env python3
"""Golden vectors for the LighterGlue oracle from an INDEPENDENT implementation of the same published algorithm:
the HuggingFace port `transformers/models/lightglue/modeling_lightglue.py` (kornia 0.7.2, the reference's actual
dependency, is neither vendored nor installed).  The port's modules are instantiated with the reference's
configuration (d = 96, 1 head) and loaded with OUR synthetic weights mapped onto its parameter names
(kornia's fused Wqkv split into q/k/v, the shared cross-attention `to_qk` tied to q_proj and k_proj).

    python tests/golden/make_golden_lighterglue.py        # writes tests/golden/lg_*.npz
"""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures  # noqa: E402
from transformers.models.lightglue.configuration_lightglue import LightGlueConfig  # noqa: E402
from transformers.models.lightglue import modeling_lightglue as M  # noqa: E402


def hf_layer(cfg, sd, i):
    layer = M.LightGlueTransformerLayer(cfg, i).eval()
    ps, pc = f"transformers.{i}.self_attn.", f"transformers.{i}.cross_attn."
    w, b = sd[ps + "Wqkv.weight"], sd[ps + "Wqkv.bias"]
    t = {
        "self_attention.q_proj.weight": w[0::3], "self_attention.q_proj.bias": b[0::3],
        "self_attention.k_proj.weight": w[1::3], "self_attention.k_proj.bias": b[1::3],
        "self_attention.v_proj.weight": w[2::3], "self_attention.v_proj.bias": b[2::3],
        "self_attention.o_proj.weight": sd[ps + "out_proj.weight"], "self_attention.o_proj.bias": sd[ps + "out_proj.bias"],
        "self_mlp.fc1.weight": sd[ps + "ffn.0.weight"], "self_mlp.fc1.bias": sd[ps + "ffn.0.bias"],
        "self_mlp.layer_norm.weight": sd[ps + "ffn.1.weight"], "self_mlp.layer_norm.bias": sd[ps + "ffn.1.bias"],
        "self_mlp.fc2.weight": sd[ps + "ffn.3.weight"], "self_mlp.fc2.bias": sd[ps + "ffn.3.bias"],
        "cross_attention.q_proj.weight": sd[pc + "to_qk.weight"], "cross_attention.q_proj.bias": sd[pc + "to_qk.bias"],
        "cross_attention.k_proj.weight": sd[pc + "to_qk.weight"], "cross_attention.k_proj.bias": sd[pc + "to_qk.bias"],
        "cross_attention.v_proj.weight": sd[pc + "to_v.weight"], "cross_attention.v_proj.bias": sd[pc + "to_v.bias"],
        "cross_attention.o_proj.weight": sd[pc + "to_out.weight"], "cross_attention.o_proj.bias": sd[pc + "to_out.bias"],
        "cross_mlp.fc1.weight": sd[pc + "ffn.0.weight"], "cross_mlp.fc1.bias": sd[pc + "ffn.0.bias"],
        "cross_mlp.layer_norm.weight": sd[pc + "ffn.1.weight"], "cross_mlp.layer_norm.bias": sd[pc + "ffn.1.bias"],
        "cross_mlp.fc2.weight": sd[pc + "ffn.3.weight"], "cross_mlp.fc2.bias": sd[pc + "ffn.3.bias"],
    }
    layer.load_state_dict({k: v.clone() for k, v in t.items()}, strict=True)
    return layer


@torch.no_grad()
def main():
    torch.manual_seed(0)
    cfg = LightGlueConfig(descriptor_dim=96, num_hidden_layers=6, num_attention_heads=1, depth_confidence=-1.0, width_confidence=0.95,
                          filter_threshold=0.1)
    cfg._attn_implementation = "eager"
    sd = fixtures.lighterglue_state_dict(0)
    N = 53
    k0, desc0, size0, k1, desc1, size1 = fixtures.lighterglue_inputs(N, N, seed=1)

    # (1) key-point normalisation + positional encoding
    kn0 = M.normalize_keypoints(k0[None], int(size0[1]), int(size0[0]))[0]
    kn1 = M.normalize_keypoints(k1[None], int(size1[1]), int(size1[0]))[0]
    pe = M.LightGluePositionalEncoder(cfg)
    pe.load_state_dict({"projector.weight": sd["posenc.Wr.weight"].clone()})
    (cos0, sin0), = pe(kn0[None])
    (cos1, sin1), = pe(kn1[None])
    np.savez(os.path.join(HERE, "lg_posenc.npz"), k0=k0.numpy(), k1=k1.numpy(), size0=size0.numpy(), size1=size1.numpy(), kn0=kn0.numpy(),
             kn1=kn1.numpy(), cos0=cos0[0].numpy(), sin0=sin0[0].numpy(), cos1=cos1[0].numpy(), sin1=sin1[0].numpy())

    # (2) transformer layers 0 and 3 on projected descriptors (the pair is the batch of 2 the port expects)
    x0 = torch.nn.functional.linear(desc0, sd["input_proj.weight"], sd["input_proj.bias"])
    x1 = torch.nn.functional.linear(desc1, sd["input_proj.weight"], sd["input_proj.bias"])
    out = {"x0": x0.numpy(), "x1": x1.numpy()}
    cos, sin = torch.cat([cos0, cos1]), torch.cat([sin0, sin1])
    for i in (0, 3):
        y, _, _ = hf_layer(cfg, sd, i)(torch.stack([x0, x1]), (cos, sin), None)
        out[f"y0_{i}"], out[f"y1_{i}"] = y[0].numpy(), y[1].numpy()
    np.savez(os.path.join(HERE, "lg_layer.npz"), **out)

    # (3) match assignment + mutual filter (layer 5 heads) on layer-3 outputs
    ma = M.LightGlueMatchAssignmentLayer(cfg)
    ma.load_state_dict({"final_projection.weight": sd["log_assignment.5.final_proj.weight"].clone(), "final_projection.bias": sd["log_assignment.5.final_proj.bias"].clone(),
                        "matchability.weight": sd["log_assignment.5.matchability.weight"].clone(), "matchability.bias": sd["log_assignment.5.matchability.bias"].clone()})
    d0, d1 = torch.from_numpy(out["y0_3"]), torch.from_numpy(out["y1_3"])
    scores = ma(torch.stack([d0, d1]), None)
    matches, mscores = M.get_matches_from_scores(scores, 0.1)
    np.savez(os.path.join(HERE, "lg_assign.npz"), d0=d0.numpy(), d1=d1.numpy(), scores=scores[0].numpy(), matches0=matches[0].numpy(),
             mscores0=mscores[0].numpy(), matchability0=ma.get_matchability(d0[None])[0].numpy())
    print("wrote lg_posenc.npz lg_layer.npz lg_assign.npz;", int((matches[0] > -1).sum()), "matches of", N)

    # (4) END TO END without width pruning: the port's own LightGlueForKeypointMatching._match_image_pair (input projection,
    #     positional encoding, 6 layers, assignment of the last layer, mutual filter) with every weight tied to ours.
    #     (Its pruning path is not used: it needs early stopping enabled and also prunes after the last layer, which is not
    #     what kornia does; pruning stays pinned only through get_matchability in (3).)
    from transformers import SuperPointConfig
    cfg2 = LightGlueConfig(keypoint_detector_config=SuperPointConfig(descriptor_decoder_dim=64), descriptor_dim=96, num_hidden_layers=6,
                           num_attention_heads=1, depth_confidence=-1.0, width_confidence=-1.0, filter_threshold=0.05)
    cfg2._attn_implementation = "eager"
    model = M.LightGlueForKeypointMatching(cfg2).eval()
    model.input_projection.load_state_dict({"weight": sd["input_proj.weight"].clone(), "bias": sd["input_proj.bias"].clone()})
    model.positional_encoder.load_state_dict({"projector.weight": sd["posenc.Wr.weight"].clone()})
    for i in range(6):
        model.transformer_layers[i].load_state_dict(hf_layer(cfg2, sd, i).state_dict())
        model.match_assignment_layers[i].load_state_dict({
            "final_projection.weight": sd[f"log_assignment.{i}.final_proj.weight"].clone(), "final_projection.bias": sd[f"log_assignment.{i}.final_proj.bias"].clone(),
            "matchability.weight": sd[f"log_assignment.{i}.matchability.weight"].clone(), "matchability.bias": sd[f"log_assignment.{i}.matchability.bias"].clone()})
    N2 = 211
    k0, desc0, size0, k1, desc1, size1 = fixtures.lighterglue_inputs(N2, N2, seed=4, size0=(640, 480), size1=(640, 480))
    m, ms, _, _, _ = model._match_image_pair(torch.stack([k0, k1])[None], torch.stack([desc0, desc1])[None], int(size0[1]), int(size0[0]),
                                             mask=torch.ones((1, 2, N2), dtype=torch.int))
    m, ms = m.reshape(2, N2), ms.reshape(2, N2)
    np.savez(os.path.join(HERE, "lg_e2e.npz"), k0=k0.numpy(), desc0=desc0.numpy(), size0=size0.numpy(), k1=k1.numpy(), desc1=desc1.numpy(),
             size1=size1.numpy(), matches0=m[0].numpy(), matches1=m[1].numpy(), mscores0=ms[0].numpy(), mscores1=ms[1].numpy(), threshold=np.float32(0.05))
    print("wrote lg_e2e.npz;", int((m[0] > -1).sum()), "matches of", N2)


if __name__ == "__main__":
    main()
