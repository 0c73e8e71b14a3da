"""Training step around `GCN_Detection_Network_extended.forward` -- the second caller of the hot path
(`/root/reference/Code/train_GENIE_model.py:1716-1861`).

Kept from the reference: the call `out = mz(*input_tensors)` with the 22 positional tensors (`:1770-1786`), the 4-term
weighted MSE `0.1 MSE(y) + 0.4 MSE(x) + 0.25 MSE(arv_p) + 0.25 MSE(arv_s)` divided by the number of valid samples of the batch
(`:1392, :1789`), `loss.backward()` per sample (`:1843-1846`) and ONE `optimizer.step()` per batch (`:1861`; Adam, lr 1e-3,
`:1383`). Not kept: the synthetic-event generator, label construction and plotting around it (out of scope).
"""
import torch

LOSS_WEIGHTS = (0.1, 0.4, 0.25, 0.25)          # train_GENIE_model.py:1392


def reference_loss(out, labels, n_valid_samples=1):
    """`(w0 MSE(out[0][:,:,0], Lbls) + w1 MSE(out[1][:,:,0], Lbls_query) + w2 MSE(out[2][:,:,0], pick_lbls[:,:,0]) +
    w3 MSE(out[3][:,:,0], pick_lbls[:,:,1])) / n_valid_samples` (train_GENIE_model.py:1789). `labels` = (Lbls [G, T],
    Lbls_query [Q, T], pick_lbls [n_src, n_picks, 2])."""
    mse = torch.nn.functional.mse_loss
    lbl, lbl_q, pick = labels
    w = LOSS_WEIGHTS
    return (w[0] * mse(out[0][:, :, 0], lbl) + w[1] * mse(out[1][:, :, 0], lbl_q) + w[2] * mse(out[2][:, :, 0], pick[:, :, 0])
            + w[3] * mse(out[3][:, :, 0], pick[:, :, 1])) / n_valid_samples


def make_optimizer(net):
    return torch.optim.Adam(net.parameters(), lr=0.001)      # train_GENIE_model.py:1383


def train_step(net, optimizer, batch):
    """One iteration of the reference's training loop over `batch` = list of (input_tensors [22], labels): zero the gradients,
    forward + loss + backward per sample, one optimizer step. Returns the summed loss value (`losses[i]`, :1862)."""
    optimizer.zero_grad()
    total = 0.0
    n = len(batch)
    for inputs, labels in batch:
        out = net(*inputs)                                    # :1786
        loss = reference_loss(out, labels, n)
        loss.backward()                                       # :1843-1846
        total += float(loss.item())
    optimizer.step()                                          # :1861
    return total
