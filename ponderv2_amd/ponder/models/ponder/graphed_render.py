"""HIP-graph replay of the render head (forward + backward) of a training step.

Once the projected volume exists, everything the NeuS head does has STATIC shapes: B scenes,
R rays per scene, 96 + 36 samples per ray, a (B,C,Z,Y,X) volume.  The eager path spends more host
time enqueuing its ~1000 small launches (samplers, MLP heads, first- and second-order sampler
gradients, compositing, six losses, and the whole backward) than the GPU spends running them.
This module captures that region ONCE into a hipGraph - the forward, the losses and
``autograd.grad`` of the loss w.r.t. the volume and the renderer's parameters - and replays it
every step.  It is exposed to autograd as a Function ``(volume, *params) -> (loss, *logged
scalars)`` whose backward hands back the gradients the graph already produced, so the upstream
part of the model (projection U-Net, sparse backbone) and DDP's gradient hooks see nothing unusual.

Exactness: the replay runs the same kernels on the same data as the eager path (the total loss is
the plain sum of the head's loss terms, so d(total)/d(head loss) == 1 and pre-computing the
backward inside the graph is exact).  Random jitter comes from the graph-safe philox generator.
"""
import torch

from .render_utils import RayBundle


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, volume, ray_dict, *params):
        outs = head._run(volume, ray_dict)
        ctx.head = head
        ctx.n_params = len(params)
        total = outs[0].clone()
        extras = [o.clone() for o in outs[1:]]
        ctx.mark_non_differentiable(*extras)
        return (total, *extras)

    @staticmethod
    def backward(ctx, g_total, *unused):
        head = ctx.head
        grads = head.static_grads
        g_vol = grads[0] * g_total if grads[0] is not None else None
        g_params = [None if g is None else g * g_total for g in grads[1:]]
        return (None, g_vol, None, *g_params)


class GraphedRenderHead:
    """Captures ``renderer(rays, volume) -> losses`` + its gradients for one (shape, dtype) key."""

    def __init__(self, model):
        self.model = model
        self.params = [p for p in model.renderer.parameters() if p.requires_grad]
        self.key = None
        self.graph = None
        self.failed = False

    # -------------------------------------------------------------- capture
    def _compute(self):
        m = self.model
        vol = self.s_volume.detach().requires_grad_(True)
        B, R = self.s_ray["ray_o"].shape[:2]
        bundle = RayBundle(origins=self.s_ray["ray_o"].reshape(B * R, 3),
                           directions=self.s_ray["ray_d"].reshape(B * R, 3), num_scenes=B)
        out = m.renderer(bundle, [vol])
        loss, loss_dict = m.render_loss(out, self.s_ray)
        grads = torch.autograd.grad(loss, [vol] + self.params, allow_unused=True)
        return loss, loss_dict, grads

    def _capture(self, volume, ray_dict):
        self.s_volume = volume.detach().clone(memory_format=torch.preserve_format)
        self.s_ray = {k: v.detach().clone() for k, v in ray_dict.items()}
        for p in self.params:
            p.view_as(p)  # bind the parameters' AccumulateGrad nodes to the caller's stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a side stream, as graph capture requires
            for _ in range(2):
                self._compute()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            loss, loss_dict, grads = self._compute()
        self.loss_names = list(loss_dict.keys())
        self.static_outs = [loss] + [loss_dict[k] for k in self.loss_names]
        self.static_grads = list(grads)

    def _run(self, volume, ray_dict):
        self.s_volume.copy_(volume.detach())
        for k, v in ray_dict.items():
            self.s_ray[k].copy_(v)
        self.graph.replay()
        return self.static_outs

    # -------------------------------------------------------------- public
    def __call__(self, volume, ray_dict):
        """-> (loss, loss_dict) like ``render_func`` + ``render_loss``; None if capture failed."""
        key = (tuple(volume.shape), volume.stride(), volume.dtype,
               tuple((k, tuple(v.shape)) for k, v in sorted(ray_dict.items())))
        if key != self.key:
            try:
                self._capture(volume, ray_dict)
                self.key = key
            except Exception as e:  # eager GPU path still works; say why the graph was refused
                import warnings

                warnings.warn(f"render-head graph capture failed, using the eager path: {e!r}")
                self.failed, self.graph, self.key = True, None, None
                return None
        res = _Replay.apply(self, volume, ray_dict, *self.params)
        return res[0], dict(zip(self.loss_names, res[1:]))
