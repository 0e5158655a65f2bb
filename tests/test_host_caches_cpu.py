"""CPU: the host-side caches of scade_amd.run_nerf_helpers.NeRF that the eager train step leans on (no kernels run).

``ordered_params()`` returns the 24 parameters in kernel order from a cache that is validated through the modules'
own dicts (round 4: the attribute walk was 60 us of an eager step); it must notice a re-assigned layer and a
re-assigned Parameter, and ``pack_key()`` must notice an in-place update (``_version``) and ``ops.PARAM_EPOCH``."""
import torch

from scade_amd import ops
from scade_amd import run_nerf_helpers as H


def _net():
    torch.manual_seed(0)
    return H.NeRF(D=8, W=256, input_ch=57, output_ch=5, skips=[4], input_ch_views=3, input_ch_cam=0, use_viewdirs=True)


def test_ordered_params_is_the_state_dict_in_kernel_order_and_cached():
    net = _net()
    ps = net.ordered_params()
    sd = dict(net.named_parameters())
    assert len(ps) == 24 and all(p is sd[k] for p, k in zip(ps, ops.PARAM_ORDER))
    assert net.ordered_params() is ps                      # the cached list object, not a new walk


def test_ordered_params_notices_reassigned_layers_and_parameters():
    net = _net()
    ps = net.ordered_params()
    # a re-assigned Parameter of the LAST layer
    net.rgb_linear.bias = torch.nn.Parameter(torch.zeros(3))
    ps2 = net.ordered_params()
    assert ps2 is not ps and ps2[-1] is net.rgb_linear.bias
    # a re-assigned Parameter of the FIRST layer
    net.pts_linears[0].weight = torch.nn.Parameter(torch.zeros(256, 57))
    ps3 = net.ordered_params()
    assert ps3 is not ps2 and ps3[0] is net.pts_linears[0].weight
    # a whole layer module replaced (its old Parameter objects live on in the old module)
    old = net.pts_linears[0]
    net.pts_linears[0] = type(old)(57, 256, activation="relu") if hasattr(old, "activation") else torch.nn.Linear(57, 256)
    ps4 = net.ordered_params()
    assert ps4 is not ps3 and ps4[0] is net.pts_linears[0].weight
    sd = dict(net.named_parameters())
    assert all(p is sd[k] for p, k in zip(ps4, ops.PARAM_ORDER))


def test_pack_key_follows_in_place_updates_and_the_raw_pointer_epoch():
    net = _net()
    k0 = net.pack_key()
    assert net.pack_key() == k0
    with torch.no_grad():
        net.pts_linears[3].bias.add_(1.0)                  # torch-visible in-place update: _version moves
    k1 = net.pack_key()
    assert k1 != k0
    before = ops.PARAM_EPOCH
    try:
        ops.PARAM_EPOCH = before + 1                       # what FusedAdam does after updating through raw pointers
        assert net.pack_key() != k1
    finally:
        ops.PARAM_EPOCH = before
