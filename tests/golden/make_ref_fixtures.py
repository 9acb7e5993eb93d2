#!/usr/bin/env python3
"""Golden fixtures from the REFERENCE'S OWN CODE for the rows next to the loss.

Run in the build container only (the reference cannot travel to the GPU box):

    python tests/golden/make_ref_fixtures.py          # writes tests/golden/ref_*.npz

It imports /root/reference/src/rnnt/model.py as it stands.  Two imports of that file's dependency chain
are absent from the image and unrelated to what is captured here: ``editdistance`` (CER scoring,
src/utils/utils.py:3) and ``kaldiio`` (feature IO, src/data/utils.py:3); both get an empty module in
``sys.modules``.  ``from warprnnt_pytorch import RNNTLoss`` (model.py:5) resolves to this repository's shim;
the loss is constructed (model.py:87) but never called here.

What is captured (inputs, parameters and the reference's outputs, nothing of its source):

* ``ref_joint_concat.npz`` / ``ref_joint_sum.npz`` — ``JointNet.forward`` (model.py:20-63) on seeded states
  and parameters, and the gradients autograd gives for every input and parameter;
* ``ref_recognize.npz`` — ``Transducer.recognize`` (model.py:135-172, the greedy loop :145-165) on a seeded
  tiny transducer: the encoder's outputs, the decoder's and the joint's parameters and the token lists the
  reference returns.  The script also runs this repository's restatement of that loop
  (oracle/decode_oracle.py) on the same modules and refuses to write the fixture unless the tokens agree.

The batched greedy search of beam_searcher.py:65-127 cannot be captured: as shipped it passes a third
argument to the joint that makes the reference's own JointNet raise (model.py:60-61).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def import_reference():
    sys.path.insert(0, ROOT)       # warprnnt_pytorch shim, oracle
    sys.path.insert(0, REFERENCE)
    for absent in ("editdistance", "kaldiio"):
        sys.modules.setdefault(absent, types.ModuleType(absent))
    import src.rnnt.model as ref_model          # noqa: E402
    from src.utils.utils import AttrDict        # noqa: E402

    return ref_model, AttrDict


def params_of(module, prefix):
    return {f"{prefix}.{k}": v.detach().numpy().copy() for k, v in module.state_dict().items()}


def joint_fixture(ref_model, kind):
    torch.manual_seed(20 if kind == "concat" else 21)
    B, T, U1, He, inner, V = 2, 5, 4, 6, 16, 11
    Hd = He                                     # 'sum' adds the two states: equal widths
    joint = ref_model.JointNet(He + Hd if kind == "concat" else He, inner, V, joint=kind).double()
    enc = torch.randn(B, T, He, dtype=torch.float64, requires_grad=True)
    dec = torch.randn(B, U1, Hd, dtype=torch.float64, requires_grad=True)
    go = torch.randn(B, T, U1, V, dtype=torch.float64)
    logits = joint(enc, dec)                    # model.py:115: no softmax
    assert logits.shape == (B, T, U1, V)
    (logits * go).sum().backward()
    out = dict(kind=np.array(kind), enc=enc.detach().numpy(), dec=dec.detach().numpy(), go=go.numpy(),
               logits=logits.detach().numpy(), grad_enc=enc.grad.numpy(), grad_dec=dec.grad.numpy())
    out.update(params_of(joint, "param"))
    for name, p in joint.named_parameters():
        out[f"grad.{name}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, f"ref_joint_{kind}.npz"), **out)
    print(f"ref_joint_{kind}.npz: logits {logits.shape}, {len(out)} arrays")


def recognize_fixture(ref_model, AttrDict):
    from oracle.decode_oracle import greedy_decode_reference

    torch.manual_seed(44)
    V, feat = 23, 9
    config = AttrDict(dict(
        feature_dim=feat, dropout=0.0, vocab_size=V, share_weight=False,
        enc=AttrDict(dict(type="lstm", hidden_size=12, output_size=10, n_layers=1, bidirectional=True)),
        dec=AttrDict(dict(type="lstm", hidden_size=14, output_size=10, n_layers=1)),
        joint=AttrDict(dict(input_size=20, inner_size=24)),
    ))
    model = ref_model.Transducer(config).eval()
    with torch.no_grad():       # scaled so that blank wins on about half the frames and argmax margins are healthy
        model.joint.mlp[2].bias[0] += 3.0
        model.joint.mlp[2].weight *= 12.0
        model.encoder.output_proj.weight *= 3
    B, T = 3, 40
    inputs = torch.randn(B, T, feat) * 4.0
    lengths = torch.tensor([40, 33, 21])         # sorted, as the reference's packed encoder expects
    with torch.no_grad():
        tokens = model.recognize(inputs, lengths)            # THE REFERENCE'S OUTPUT
        enc_states, out_lengths = model.encoder(inputs, lengths)
    assert any(len(t) > 0 for t in tokens) and all(len(t) < int(n) for t, n in zip(tokens, lengths))
    margins = []
    for i in range(B):                           # the restatement must reproduce it on the same modules
        mine, margin = greedy_decode_reference(enc_states[i], int(out_lengths[i]), model.decoder,
                                               lambda e, d: model.joint(e, d))
        assert mine == tokens[i], (i, mine, tokens[i])
        margins.append(margin)
    flat = np.array([t for seq in tokens for t in seq], dtype=np.int64)
    offsets = np.cumsum([0] + [len(seq) for seq in tokens]).astype(np.int64)
    out = dict(enc_states=enc_states.numpy(), out_lengths=out_lengths.numpy().astype(np.int64), tokens=flat,
               token_offsets=offsets, min_top2_margin=np.array(min(margins)), vocab=np.array(V),
               dec_hidden=np.array(14), dec_out=np.array(10), joint_inner=np.array(24))
    out.update(params_of(model.decoder, "decoder"))
    out.update(params_of(model.joint, "joint"))
    np.savez_compressed(os.path.join(HERE, "ref_recognize.npz"), **out)
    print("ref_recognize.npz: tokens", [len(t) for t in tokens], "min top-2 margin %.3g" % min(margins))


def main():
    ref_model, AttrDict = import_reference()
    joint_fixture(ref_model, "concat")
    joint_fixture(ref_model, "sum")
    recognize_fixture(ref_model, AttrDict)


if __name__ == "__main__":
    main()
