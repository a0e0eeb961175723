"""detzero_det.models (detection/detzero_det/models/__init__.py:1-48)."""
from collections import namedtuple

from detzero_amd.centerpoint import CenterPoint, __all__, build_network, load_data_to_gpu  # noqa: F401
from detzero_amd.lib import DetZeroHipError


def model_fn_decorator():
    """models/__init__.py:31-46 - the training step wrapper; training is out of scope of the HIP backend."""
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])      # noqa: F841  (name kept)

    def model_func(model, batch_dict):
        raise DetZeroHipError('model_fn_decorator: training is out of scope of the HIP backend')
    return model_func
