"""Training-mode forward of the DIGAT encoder (autograd through the HIP path).

Not implemented in this revision: the backward kernels (digat_xattn_bwd & co., SURVEY.md §8b item
2) are the next row of the scope table.  Failing loudly here is deliberate — there is no eager
PyTorch fallback that could silently stand in for the native path.
"""


def digat_forward_train(encoder, *inputs):
    raise NotImplementedError(
        "digat_amd: training-mode forward/backward through the HIP kernels is not implemented yet; "
        "use model.eval() / torch.no_grad() for the inference path")
