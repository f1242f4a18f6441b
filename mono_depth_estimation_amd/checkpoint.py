"""Checkpoint bridge: Lightning-style checkpoints <-> the drop-in FCRN module.

The reference saves through ``pl.callbacks.ModelCheckpoint`` (train.py:102-121) and restores with
``FCRNModule.load_from_checkpoint`` (modules/__init__.py:26-28); the network sits under the attribute
``model`` of the LightningModule (modules/laina.py:15), so its weights appear in ``ckpt["state_dict"]`` as
``"model." + key`` with exactly the keys this package's ``network.FCRN.ResNet.state_dict()`` has, in OIHW fp32.
The optimiser is ``torch.optim.Adam`` over two parameter groups, encoder then decoder (laina.py:51-57); its
per-parameter moments map onto the engine's flat moment buffers (conv weights are stored OHWI there).

Loading a checkpoint unpickles it: only load files you trust (Lightning checkpoints carry argparse
namespaces, so ``weights_only`` loading is not possible for them).
"""
import collections

import torch

PREFIX = "model."


def model_state_from_checkpoint(ckpt, prefix=PREFIX):
    """The network's state_dict out of a Lightning checkpoint dict (or a plain state_dict, returned as is)."""
    sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    if any(k.startswith(prefix) for k in sd):
        return collections.OrderedDict((k[len(prefix):], v) for k, v in sd.items() if k.startswith(prefix))
    return collections.OrderedDict(sd)


def load_checkpoint(module, path_or_ckpt, strict=True, map_location="cpu"):
    """Load network weights and BatchNorm buffers into `module`; returns the checkpoint dict (epoch,
    global_step, optimizer_states ... for the caller).  Works before or after the module's first forward:
    parameters are updated in place, the engine re-derives its bf16 / transposed packings on the next step."""
    ckpt = torch.load(path_or_ckpt, map_location=map_location, weights_only=False) if isinstance(path_or_ckpt, str) else path_or_ckpt
    missing, unexpected = module.load_state_dict(model_state_from_checkpoint(ckpt), strict=strict)
    if not strict and (missing or unexpected):
        import warnings
        warnings.warn("checkpoint: %d missing and %d unexpected keys" % (len(missing), len(unexpected)))
    return ckpt if isinstance(ckpt, dict) else {"state_dict": ckpt}


def save_checkpoint(module, path, epoch=0, global_step=0, optimizer_states=None, hyper_parameters=None, prefix=PREFIX):
    """Write a checkpoint with the layout the reference's Lightning 1.4 run produces for the network:
    ``state_dict`` (prefixed, OIHW fp32 on CPU), ``epoch``, ``global_step``, optional ``optimizer_states``."""
    sd = collections.OrderedDict((prefix + k, v.detach().to("cpu").contiguous().clone()) for k, v in module.state_dict().items())
    ckpt = {"epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": "1.4.0", "state_dict": sd}
    if optimizer_states is not None:
        ckpt["optimizer_states"] = optimizer_states
    if hyper_parameters is not None:
        ckpt["hyper_parameters"] = hyper_parameters
    torch.save(ckpt, path)
    return ckpt


def _param_groups(module):
    """Parameters in the reference optimiser's order: get_1x_lr_params (encoder), get_10x_lr_params (decoder)."""
    return [list(module.get_1x_lr_params()), list(module.get_10x_lr_params())]


def _oihw(store, flat, p):
    return store.view_of(flat, p)


def adam_state_dict(module, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam.state_dict() equivalent of the engine's flat Adam state (module must have run a step)."""
    store = module._store
    if store is None or store.adam_state is None:
        raise RuntimeError("adam_state_dict: the module has not taken an optimiser step yet")
    mom, var = store.adam_state
    groups, state, idx = _param_groups(module), {}, 0
    pg = []
    for gi, params in enumerate(groups):
        ids = []
        for p in params:
            state[idx] = {"step": torch.tensor(float(store.step_count)),
                          "exp_avg": _oihw(store, mom, p).detach().to("cpu").contiguous().clone(),
                          "exp_avg_sq": _oihw(store, var, p).detach().to("cpu").contiguous().clone()}
            ids.append(idx)
            idx += 1
        pg.append({"lr": lr * (1 if gi == 0 else 10), "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay,
                   "amsgrad": False, "params": ids})
    return {"state": state, "param_groups": pg}


def load_adam_state_dict(module, sd):
    """Inverse of adam_state_dict: fill the engine's flat moments (and step count) from a torch.optim.Adam
    state_dict whose parameter order is the reference's (encoder group, then decoder group)."""
    store = module._store
    if store is None:
        raise RuntimeError("load_adam_state_dict: run one forward first so the flat parameter store exists")
    if store.adam_state is None:
        store.adam_state = (torch.zeros_like(store.P), torch.zeros_like(store.P))
    mom, var = store.adam_state
    params = [p for g in _param_groups(module) for p in g]
    order = [i for g in sd["param_groups"] for i in g["params"]]
    if len(order) != len(params):
        raise ValueError("optimizer state has %d parameters, the network %d" % (len(order), len(params)))
    steps = set()
    with torch.no_grad():
        for p, i in zip(params, order):
            st = sd["state"].get(i)
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state %d has shape %s, parameter %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            _oihw(store, mom, p).copy_(st["exp_avg"].to(mom.device))
            _oihw(store, var, p).copy_(st["exp_avg_sq"].to(var.device))
            steps.add(int(st["step"]))
    if len(steps) > 1:
        raise ValueError("optimizer state carries different step counts per parameter: %s" % sorted(steps))
    if steps:
        store.step_count = steps.pop()


# ---------------------------------------------------------------------------------------------- pretrained-trunk bridges (N3)
def vnl_resnext_keys(src_dict):
    """reference network/VNL.py:44-67 (`convert_state_dict_resnext`): the ResNeXt ImageNet files the VNL authors ship
    (`pretrained_models/ResNeXt-ImageNet/resnext{50,101}_32x4d.pth`) are nn.Sequential dumps of the original Torch7 model,
    keys are index paths.  Top level: 0 = stem conv, 1 = stem BN, 4..7 = the four stages; inside a block the residual
    branch is `<stage>.<block>.0.0.<k>`, itself a Sequential for the first two conv/BN pairs (one more index: 0 conv1,
    1 bn1, 3 conv2, 4 bn2 -> 7-token keys) followed by conv3 (1) and bn3 (2), and the projection shortcut is
    `<stage>.<block>.0.1.<k>` (0 conv, 1 bn) -> 6-token keys.  Returns {body key ('res2.0.conv1.weight', ...): tensor};
    everything else (classifier, pooling) is dropped."""
    inner = {0: "conv1.", 1: "bn1.", 3: "conv2.", 4: "bn2."}
    outer = ({1: "conv3.", 2: "bn3."}, {0: "shortcut.conv.", 1: "shortcut.bn."})
    out = {}
    for k, v in src_dict.items():
        t = k.split(".")
        top = int(t[0])
        if top == 0:
            out["res1.conv1." + t[-1]] = v
        elif top == 1:
            out["res1.bn1." + t[-1]] = v
        elif 4 <= top <= 7:
            head = "res%d.%d." % (top - 2, int(t[1]))
            if len(t) == 7:
                out[head + inner[int(t[-2])] + t[-1]] = v
            elif len(t) == 6:
                out[head + outer[int(t[-3])][int(t[-2])] + t[-1]] = v
    return out


def load_vnl_imagenet_weights(model, path_or_dict, map_location="cpu"):
    """reference VNL.py:70-95 (`load_pretrained_imagenet_weights`) for the ResNeXt encoders: copy every converted tensor
    whose key exists in `model.depth_model.encoder_modules.bottomup` (a MetricDepthModel), report the others.  The
    reference resolves the file under <cwd>/mono-depth-estimation/network/pretrained_models/...; here the caller names it."""
    src = torch.load(path_or_dict, map_location=map_location) if isinstance(path_or_dict, str) else path_or_dict
    body = model.depth_model.encoder_modules.bottomup
    own = body.state_dict()
    unknown = []
    with torch.no_grad():
        for k, v in vnl_resnext_keys(src).items():
            if k in own:
                own[k].copy_(v)
            else:
                unknown.append(k)
    return unknown


def load_midas_weights(model, path, map_location="cpu"):
    """reference MiDaS.py:10-23 (`BaseModel.load`): a state_dict file, or a training checkpoint holding it under "model"."""
    parameters = torch.load(path, map_location=map_location)
    if "optimizer" in parameters:
        parameters = parameters["model"]
    model.load_state_dict(parameters)
