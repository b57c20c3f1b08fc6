"""Build-container stub (NOT reference code): the minimum of torchvision the reference imports, so that
/root/reference can be imported here to generate golden vectors.  roi_pool is the oracle's restatement."""
from . import ops, utils, transforms  # noqa: F401
