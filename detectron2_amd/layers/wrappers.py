"""Small helpers the hot-path wrappers need (reference: detectron2/layers/wrappers.py:51-62,
:150-163)."""
import torch


def disable_torch_compiler(func):
    """Keep `func` opaque to torch.compile (reference: layers/wrappers.py:51-62).  torch.compiler.disable wraps
    every call in a dynamo frame guard (~20 us, more than some of the kernels behind these ops), so the wrapper
    is only entered while a compile is actually tracing; eager calls go straight to `func`."""
    if not (hasattr(torch, "compiler") and hasattr(torch.compiler, "disable")):
        return func
    disabled = torch.compiler.disable(func)
    is_compiling = getattr(torch.compiler, "is_compiling", None)
    if is_compiling is None:
        return disabled

    import functools

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        if is_compiling():
            return disabled(*args, **kwargs)
        return func(*args, **kwargs)

    return wrapper


class _NewEmptyTensorOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, new_shape):
        ctx.shape = x.shape
        return x.new_empty(new_shape)

    @staticmethod
    def backward(ctx, grad):
        shape = ctx.shape
        return _NewEmptyTensorOp.apply(grad, shape), None
