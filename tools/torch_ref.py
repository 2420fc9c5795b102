"""Independent second implementation of the forward pass on torch-CPU (dev-time checker only).

Used by tests/test_oracle.py and tools/make_nn_golden.py to cross-check oracle/model_np.py:
torch.nn.LSTM (gate order i, f, g, o; weights [4H, in]) is fed the TF-layout kernels
([in+H, 4H], gate order i, c~, f, o -- clair/model.py:301 CudnnCompatibleLSTMCell) after the
gate permutation, and torch.selu / softmax replace the hand-written ones.  Never imported by
the product.
"""
import numpy as np
import torch

H = 128


def _lstm_from_tf(kernel_fw, bias_fw, kernel_bw, bias_bw, in_dim, dtype):
    lstm = torch.nn.LSTM(in_dim, H, num_layers=1, bidirectional=True).to(dtype)
    perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
    with torch.no_grad():
        for sfx, k, b in (("", kernel_fw, bias_fw), ("_reverse", kernel_bw, bias_bw)):
            k = np.asarray(k)[:, perm]
            b = np.asarray(b)[perm]
            getattr(lstm, "weight_ih_l0" + sfx).copy_(torch.from_numpy(k[:in_dim].T.copy()).to(dtype))
            getattr(lstm, "weight_hh_l0" + sfx).copy_(torch.from_numpy(k[in_dim:].T.copy()).to(dtype))
            getattr(lstm, "bias_ih_l0" + sfx).copy_(torch.from_numpy(b.copy()).to(dtype))
            getattr(lstm, "bias_hh_l0" + sfx).zero_()
    return lstm


def forward(w, x, dtype=torch.float32):
    torch.set_num_threads(4)
    n = x.shape[0]
    with torch.no_grad():
        s = torch.from_numpy(np.ascontiguousarray(x)).to(dtype).reshape(n, 33, 32).transpose(0, 1)  # [T,n,32]
        l1 = _lstm_from_tf(w["lstm1_fw_kernel"], w["lstm1_fw_bias"], w["lstm1_bw_kernel"], w["lstm1_bw_bias"], 32, dtype)
        l2 = _lstm_from_tf(w["lstm2_fw_kernel"], w["lstm2_fw_bias"], w["lstm2_bw_kernel"], w["lstm2_bw_bias"], 256, dtype)
        a1, _ = l1(s)
        a2, _ = l2(a1)
        a2b = a2.transpose(0, 1)                                         # [n,T,256]
        w3 = torch.from_numpy(w["l3_kernel"]).to(dtype)                   # [256,33,30]
        b3 = torch.from_numpy(w["l3_bias"]).to(dtype)                     # [256,30]
        # per-channel dense over positions -> [n,30,256]
        l3 = torch.selu(torch.einsum("ntc,ctu->nuc", a2b, w3) + b3.t().unsqueeze(0))
        v = l3.reshape(n, 7680)
        l4 = torch.selu(v @ torch.from_numpy(w["l4_kernel"]).to(dtype) + torch.from_numpy(w["l4_bias"]).to(dtype))
        outs = []
        for k, name in enumerate(("gt21", "genotype", "len1", "len2")):
            l5 = torch.selu(l4 @ torch.from_numpy(w["l5_kernel"][k]).to(dtype) + torch.from_numpy(w["l5_bias"][k]).to(dtype))
            lg = torch.selu(l5 @ torch.from_numpy(w["head_%s_kernel" % name]).to(dtype)
                            + torch.from_numpy(w["head_%s_bias" % name]).to(dtype))
            outs.append(torch.softmax(lg, dim=1).numpy())
        return outs, dict(a1=a1.numpy(), a2=a2.numpy(), l3=l3.numpy(), l4=l4.numpy())
