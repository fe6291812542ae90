"""modelcompose/utils.py of the reference, the part the eval / demo callers use."""


def disable_torch_init():
    """utils.py:93-99 skips torch.nn default initialisers to speed up model construction; the HIP path builds no torch.nn
    modules (weights go from the checkpoint straight into packed HBM buffers), so there is nothing to disable."""
    return None
