"""Drop-in for reference network/FCRN.py on MI355X.

Same public surface as the reference file — ``ResNet(dataset, layers, decoder, output_size,
in_channels, out_channels, pretrained)`` with ``forward``, ``get_1x_lr_params``,
``get_10x_lr_params``, ``weights_init`` and byte-identical ``state_dict`` keys (397 for
ResNet-50) — but ``forward``/``backward`` run on hand-written gfx950 HIP
kernels through libmde_hip.so (mono_depth_estimation_amd/engine.py).  The submodules below
are parameter containers only: they keep the reference's names so checkpoints load, and have
no arithmetic of their own (there is no PyTorch fallback path).
"""
import collections
import copy
import math
import os
import warnings

import torch
import torch.nn as nn

from .. import _lib
from ..engine import FCRNEngine, ParamStore


def weights_init(m):
    """reference network/FCRN.py:14-28 — He-normal with fan = kh*kw*Cout; BN -> (1, 0)."""
    if isinstance(m, nn.Conv2d):
        n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
        m.weight.data.normal_(0, math.sqrt(2. / n))
        if m.bias is not None:
            m.bias.data.zero_()
    elif isinstance(m, nn.ConvTranspose2d):
        n = m.kernel_size[0] * m.kernel_size[1] * m.in_channels
        m.weight.data.normal_(0, math.sqrt(2. / n))
        if m.bias is not None:
            m.bias.data.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        m.weight.data.fill_(1)
        m.bias.data.zero_()


class _Container(nn.Module):
    """A module that only holds parameters; the HIP engine does the arithmetic."""

    def forward(self, *a, **k):
        raise RuntimeError("%s is a parameter container of the HIP FCRN path; call the parent "
                           "mono_depth_estimation_amd.network.FCRN.ResNet instead" % type(self).__name__)


class _Seq(nn.Sequential):
    """nn.Sequential as a parameter container (same child names -> same state_dict keys); never run by torch."""

    def forward(self, *a, **k):
        raise RuntimeError("this Sequential is a parameter container of the HIP FCRN path; call the parent "
                           "mono_depth_estimation_amd.network.FCRN.ResNet instead")


class Bottleneck(_Container):
    """torchvision Bottleneck (v1.5: stride on conv2) — names as torchvision's."""
    expansion = 4

    def __init__(self, cin, width, stride=1, project=False):
        super().__init__()
        cout = width * 4
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))


class BasicBlock(_Container):
    """torchvision BasicBlock (resnet18 / resnet34) — names as torchvision's."""
    expansion = 1

    def __init__(self, cin, width, stride=1, project=False):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(width, width, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, width, 1, stride=stride, bias=False), nn.BatchNorm2d(width))


_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def _stage(cin, width, n, stride, basic=False):
    if basic:
        mods = [BasicBlock(cin, width, stride, project=(stride != 1 or cin != width))]
        mods += [BasicBlock(width, width) for _ in range(1, n)]
        return nn.Sequential(*mods)
    mods = [Bottleneck(cin, width, stride, project=True)]
    mods += [Bottleneck(width * 4, width) for _ in range(1, n)]
    return nn.Sequential(*mods)


class Unpool(_Container):
    """reference FCRN.py:31-44.  Never materialised on the HIP path: the zero insertion is
    folded into the following 5x5 convs as four output phases (csrc/conv_gemm.hip)."""

    def __init__(self, num_channels, stride=2):
        super().__init__()
        self.num_channels, self.stride = num_channels, stride


class Decoder(_Container):
    names = ['deconv2', 'deconv3', 'upconv', 'upproj']


class DeConv(Decoder):
    """reference FCRN.py:68-88: four x {ConvTranspose2d(k, 2, (k-1)//2, k%2, bias=False) -> BN -> ReLU}."""
    kind = "deconv"

    def __init__(self, in_channels, kernel_size):
        assert kernel_size >= 2, "kernel_size out of range: {}".format(kernel_size)
        super().__init__()
        if kernel_size > 5:
            raise NotImplementedError("HIP deconv decoder: kernel_size <= 5 (%d taps per launch)" % (kernel_size ** 2))
        self.kernel_size = kernel_size

        def convt(in_channels):
            stride = 2
            padding = (kernel_size - 1) // 2
            output_padding = kernel_size % 2
            assert -2 - 2 * padding + kernel_size + output_padding == 0, "deconv parameters incorrect"
            module_name = "deconv{}".format(kernel_size)
            return _Seq(collections.OrderedDict([
                (module_name, nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size, stride, padding,
                                                 output_padding, bias=False)),
                ('batchnorm', nn.BatchNorm2d(in_channels // 2)),
                ('relu', nn.ReLU(inplace=True)),
            ]))

        self.layer1 = convt(in_channels)
        self.layer2 = convt(in_channels // 2)
        self.layer3 = convt(in_channels // (2 ** 2))
        self.layer4 = convt(in_channels // (2 ** 3))


class UpConv(Decoder):
    """reference FCRN.py:91-110: four x {unpool -> 5x5 conv -> BN -> ReLU}."""
    kind = "upconv"

    def upconv_module(self, in_channels):
        return _Seq(collections.OrderedDict([
            ('unpool', Unpool(in_channels)),
            ('conv', nn.Conv2d(in_channels, in_channels // 2, kernel_size=5, stride=1, padding=2, bias=False)),
            ('batchnorm', nn.BatchNorm2d(in_channels // 2)),
            ('relu', nn.ReLU()),
        ]))

    def __init__(self, in_channels):
        super().__init__()
        self.layer1 = self.upconv_module(in_channels)
        self.layer2 = self.upconv_module(in_channels // 2)
        self.layer3 = self.upconv_module(in_channels // 4)
        self.layer4 = self.upconv_module(in_channels // 8)


class UpProj(Decoder):
    """reference FCRN.py:167-205."""
    kind = "upproj"

    class UpProjModule(_Container):
        def __init__(self, in_channels):
            super().__init__()
            out_channels = in_channels // 2
            self.unpool = Unpool(in_channels)
            self.upper_branch = nn.Sequential(collections.OrderedDict([
                ('conv1', nn.Conv2d(in_channels, out_channels, kernel_size=5, stride=1, padding=2, bias=False)),
                ('batchnorm1', nn.BatchNorm2d(out_channels)),
                ('relu', nn.ReLU()),
                ('conv2', nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False)),
                ('batchnorm2', nn.BatchNorm2d(out_channels)),
            ]))
            self.bottom_branch = nn.Sequential(collections.OrderedDict([
                ('conv', nn.Conv2d(in_channels, out_channels, kernel_size=5, stride=1, padding=2, bias=False)),
                ('batchnorm', nn.BatchNorm2d(out_channels)),
            ]))
            self.relu = nn.ReLU()

    def __init__(self, in_channels):
        super().__init__()
        self.layer1 = self.UpProjModule(in_channels)
        self.layer2 = self.UpProjModule(in_channels // 2)
        self.layer3 = self.UpProjModule(in_channels // 4)
        self.layer4 = self.UpProjModule(in_channels // 8)


class FasterUpConv(Decoder):
    """reference FCRN.py:113-164 (pixel-shuffle formulation of UpConv; its own weights).  The reference defines the
    class but its choose_decoder never returns it; here `decoder='fasterupconv'` selects it."""
    kind = "fasterupconv"

    class faster_upconv_module(_Container):
        def __init__(self, in_channel):
            super().__init__()
            for name, ks in (("conv1_", 3), ("conv2_", (2, 3)), ("conv3_", (3, 2)), ("conv4_", 2)):
                setattr(self, name, _Seq(collections.OrderedDict([
                    ('conv1', nn.Conv2d(in_channel, in_channel // 2, kernel_size=ks)),
                    ('bn1', nn.BatchNorm2d(in_channel // 2)),
                ])))
            self.ps = nn.PixelShuffle(2)
            self.relu = nn.ReLU(inplace=True)

    def __init__(self, in_channel):
        super().__init__()
        self.layer1 = self.faster_upconv_module(in_channel)
        self.layer2 = self.faster_upconv_module(in_channel // 2)
        self.layer3 = self.faster_upconv_module(in_channel // 4)
        self.layer4 = self.faster_upconv_module(in_channel // 8)


class FasterUpProj(Decoder):
    """reference FCRN.py:206-281 (pixel-shuffle formulation; its own weights, not UpProj's)."""
    kind = "fasterupproj"

    class faster_upconv(_Container):
        def __init__(self, in_channel):
            super().__init__()
            for name, ks in (("conv1_", 3), ("conv2_", (2, 3)), ("conv3_", (3, 2)), ("conv4_", 2)):
                setattr(self, name, _Seq(collections.OrderedDict([
                    ('conv1', nn.Conv2d(in_channel, in_channel // 2, kernel_size=ks)),
                    ('bn1', nn.BatchNorm2d(in_channel // 2)),
                ])))
            self.ps = nn.PixelShuffle(2)
            self.relu = nn.ReLU(inplace=True)

    class FasterUpProjModule(_Container):
        def __init__(self, in_channels):
            super().__init__()
            out_channels = in_channels // 2
            self.upper_branch = _Seq(collections.OrderedDict([
                ('faster_upconv', FasterUpProj.faster_upconv(in_channels)),
                ('relu', nn.ReLU(inplace=True)),
                ('conv', nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False)),
                ('batchnorm', nn.BatchNorm2d(out_channels)),
            ]))
            self.bottom_branch = FasterUpProj.faster_upconv(in_channels)
            self.relu = nn.ReLU(inplace=True)

    def __init__(self, in_channel):
        super().__init__()
        self.layer1 = self.FasterUpProjModule(in_channel)
        self.layer2 = self.FasterUpProjModule(in_channel // 2)
        self.layer3 = self.FasterUpProjModule(in_channel // 4)
        self.layer4 = self.FasterUpProjModule(in_channel // 8)


def choose_decoder(decoder, in_channels):
    """reference FCRN.py:282-294: every option runs on the same conv kernels ('upproj' is the reference's default,
    the one modules/laina.py uses)."""
    if decoder[:6] == 'deconv':
        assert len(decoder) == 7
        return DeConv(in_channels, int(decoder[6]))
    if decoder == "upproj":
        return UpProj(in_channels)
    if decoder == "upconv":
        return UpConv(in_channels)
    if decoder == "fasterupproj":
        return FasterUpProj(in_channels)
    if decoder == "fasterupconv":          # (not offered by the reference's choose_decoder; the class is, FCRN.py:113)
        return FasterUpConv(in_channels)
    assert False, "invalid option for decoder: {}".format(decoder)


class _FCRNFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are the engine's plans.
    backward RETURNS the parameter gradients — fresh views of a flat gradient buffer, so autograd adopts them as
    .grad without a copy when .grad is None (optimizer.zero_grad()), or adds them onto an existing .grad that lives
    in the other flat buffer (gradient accumulation).  Returning None and writing .grad behind autograd's back would
    bypass AccumulateGrad: DistributedDataParallel (what Lightning wraps the model in) then never reduces anything."""

    @staticmethod
    def forward(ctx, x, engine, train, *params):
        ctx.engine, ctx.train = engine, train
        y = engine.forward(x, train, check_data=True)
        engine.forward_serial = ctx.serial = getattr(engine, "forward_serial", 0) + 1
        return y.clone()

    @staticmethod
    def backward(ctx, dy):
        _lib.note_fp16_backward()
        eng = ctx.engine
        if not ctx.train:
            # the backward plan is the TRAINING-mode one (batch statistics, the ReLU masks a train-mode forward wrote);
            # after an eval-mode forward those buffers hold another step's values or nothing at all
            raise RuntimeError(
                "mono_depth_estimation_amd FCRN: backward() through a forward pass run in eval() mode is not supported "
                "(BatchNorm backward with running statistics has no kernel here); call .train() before the forward "
                "pass whose gradients you need, or wrap evaluation in torch.no_grad().")
        if ctx.needs_input_grad[0]:
            raise RuntimeError("mono_depth_estimation_amd FCRN: the gradient with respect to the input image is not "
                               "computed (the stem has no input-gradient kernel); detach the input.")
        if ctx.serial != eng.forward_serial:
            # the launch plan of one input shape owns ONE set of activation buffers: a later forward of the same shape
            # has replaced what this backward needs.  Refuse instead of returning gradients of the wrong activations.
            raise RuntimeError(
                "mono_depth_estimation_amd FCRN: backward() of a forward pass whose activations were overwritten by a "
                "later forward of the same input shape (forward #%d, latest #%d).  Run forward and backward in pairs "
                "(gradient accumulation over micro-batches does), or keep a second module copy for the interleaved "
                "pass." % (ctx.serial, eng.forward_serial))
        st = eng.store
        buf = st.begin_autograd_backward()
        red = st.grad_reducer
        try:
            if red is not None:                  # the gradient exchange overlapped with this backward (ResNet.set_grad_reducer)
                red.begin(buf)
                eng.backward(dy.contiguous(), red.ready, consumer_waits_side=eng.side in red.extra_streams)
            else:
                eng.backward(dy.contiguous())
        finally:
            st.Gcur = st.G                       # the direct (non-autograd) path always accumulates into G
        grads = tuple(st.grad_view(p, buf) if need else None for p, need in zip(eng.params, ctx.needs_input_grad[3:]))
        if st._g_base is None:                   # torch cannot tell who still views the flat buffers: hand out copies
            grads = tuple(g.clone() if g is not None else None for g in grads)
        return (None, None, None) + grads


class ResNet(nn.Module):
    def __init__(self, dataset='kitti', layers=50, decoder='upproj', output_size=(228, 304), in_channels=3,
                 out_channels=20, pretrained=True):
        if layers not in [18, 34, 50, 101, 152]:
            raise RuntimeError('Only 18, 34, 50, 101, and 152 layer model are defined for ResNet. Got {}'.format(layers))
        if not 1 <= in_channels <= 64:
            raise NotImplementedError("HIP FCRN stem: 1 <= in_channels <= 64 (got %d)" % in_channels)
        super(ResNet, self).__init__()
        n = _BLOCKS[layers]
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.output_size = tuple(output_size)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        basic = layers <= 34
        e = 1 if basic else 4
        self.layer1 = _stage(64, 64, n[0], 1, basic)
        self.layer2 = _stage(64 * e, 128, n[1], 2, basic)
        self.layer3 = _stage(128 * e, 256, n[2], 2, basic)
        self.layer4 = _stage(256 * e, 512, n[3], 2, basic)
        # torchvision's default init of the trunk (pretrained=False): He fan-out normal, BN (1, 0)
        for m in (self.conv1, self.layer1, self.layer2, self.layer3, self.layer4):
            for mod in m.modules():
                if isinstance(mod, nn.Conv2d):
                    nn.init.kaiming_normal_(mod.weight, mode='fan_out', nonlinearity='relu')
        if in_channels != 3:                                   # reference FCRN.py:309-313: its own conv1 / bn1
            weights_init(self.conv1)
            weights_init(self.bn1)
        num_channels = 512 if basic else 2048           # reference FCRN.py:329-332
        self.conv2 = nn.Conv2d(num_channels, num_channels // 2, kernel_size=1, bias=False)
        self.bn2 = nn.BatchNorm2d(num_channels // 2)
        self.upSample = choose_decoder(decoder, num_channels // 2)
        self.conv3 = nn.Conv2d(num_channels // 32, out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bilinear = nn.Upsample(size=self.output_size, mode='bilinear', align_corners=True)
        self.conv2.apply(weights_init)
        self.bn2.apply(weights_init)
        self.upSample.apply(weights_init)
        self.conv3.apply(weights_init)
        if pretrained:
            self._load_pretrained_trunk(layers)
        self._engines = {}
        self._store = None

    def _load_pretrained_trunk(self, layers):
        """The reference downloads torchvision ImageNet weights (FCRN.py:305); there is no
        network here, so they are read from $MDE_PRETRAINED_RESNET (a torchvision resnetNN
        state_dict file) when set."""
        path = os.environ.get("MDE_PRETRAINED_RESNET")
        if not path:
            warnings.warn("pretrained=True but $MDE_PRETRAINED_RESNET is not set: the ResNet-%d trunk keeps its "
                          "random initialisation (the reference would download torchvision weights)" % layers)
            return
        sd = torch.load(path, map_location="cpu")
        own = self.state_dict()
        own.update({k: v for k, v in sd.items() if k in own and not k.startswith("fc.")})
        self.load_state_dict(own)

    # The flat parameter store and the per-shape launch plans are caches keyed by THIS module's Parameter objects, and
    # once they exist the Parameters / buffers are VIEWS of a few big flat tensors.  A copy must not drag those along
    # (copying or pickling a view copies its whole storage): every parameter and buffer is cloned into a standalone
    # tensor (same values, same channels_last strides) and the copy rebuilds its own caches at its first forward.
    def __deepcopy__(self, memo):
        with torch.no_grad():
            for p in self.parameters():
                memo[id(p)] = nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad)
            for b in self.buffers():
                memo[id(b)] = b.detach().clone()
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ("_engines", "_store"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        new._engines, new._store = {}, None
        return new

    def __getstate__(self):
        if self._store is None:
            state = dict(self.__dict__)
            state["_engines"] = {}
            return state
        return copy.deepcopy(self).__dict__          # standalone tensors; pickling the views would write the flat buffers per tensor

    def _apply(self, fn, *args, **kwargs):
        """.cuda() / .to(device) / .float(): after the move, build the flat parameter store right away when the
        parameters sit on a GPU.  Wrappers that record parameter layouts at construction (DistributedDataParallel
        lays out its gradient buckets with the parameters' strides) must see the final channels_last views, not
        the contiguous tensors they replace at the first forward: with mismatched layouts DDP reduced the k>1 conv
        gradients wrongly."""
        out = super()._apply(fn, *args, **kwargs)
        if getattr(self, "_engines", None) is not None:           # fully constructed
            dev = self.conv1.weight.device
            self._engines, self._store = {}, None
            if dev.type == "cuda":
                self._store = ParamStore(self, dev)
        return out

    def set_grad_reducer(self, reducer):
        """Overlap the data-parallel gradient exchange with backward on the nn.Module path: see graph.TapeModule.set_grad_reducer
        (a dp.FlatGradReducer over `self._store.G` with `engine.grad_boundaries()`; join with `reducer.finish()`)."""
        if self._store is None:
            raise RuntimeError("set_grad_reducer: move the module to its GPU first (the flat gradient buffer lives there)")
        self._store.grad_reducer = reducer

    def _engine(self, x):
        if x.dim() != 4 or x.shape[1] != self.conv1.in_channels:
            raise ValueError("expected an N x %d x H x W image batch, got %s" % (self.conv1.in_channels, tuple(x.shape)))
        if not x.is_cuda:
            raise RuntimeError("mono_depth_estimation_amd FCRN runs on MI355X only (input is on %s); there is no "
                               "CPU fallback" % x.device)
        st = self._store
        if st is None or st.dev != x.device or not st.storage_is_current():
            # first use, or the parameters were moved / replaced: (re)build the flat stores
            self._store = st = ParamStore(self, x.device)
            self._engines.clear()
        key = tuple(x.shape)
        eng = self._engines.pop(key, None)
        if eng is None:
            # a launch plan owns every activation buffer of its shape (GBs at training batch sizes): keep the most
            # recently used few ($MDE_MAX_PLANS, default 4), e.g. train + validation shapes; a plan that still has a
            # backward pending stays alive through the autograd graph that references it.
            cap = max(1, int(os.environ.get("MDE_MAX_PLANS", "4")))
            while len(self._engines) >= cap:
                self._engines.pop(next(iter(self._engines)))
            eng = FCRNEngine(self, st, x.shape[0], x.shape[2], x.shape[3])
        self._engines[key] = eng                     # (re)insert as the most recent
        return eng

    def forward(self, x):
        eng = self._engine(x)
        x = x.contiguous().float()
        return _FCRNFunction.apply(x, eng, self.training, *eng.params)

    def get_1x_lr_params(self):
        """Encoder parameters (reference FCRN.py:373-381)."""
        b = [self.conv1, self.bn1, self.relu, self.maxpool, self.layer1, self.layer2, self.layer3, self.layer4]
        for i in range(len(b)):
            for k in b[i].parameters():
                if k.requires_grad:
                    yield k

    def get_10x_lr_params(self):
        """conv2/bn2/decoder/conv3 parameters (reference FCRN.py:383-391)."""
        b = [self.conv2, self.bn2, self.upSample, self.conv3, self.bilinear]
        for j in range(len(b)):
            for k in b[j].parameters():
                if k.requires_grad:
                    yield k
