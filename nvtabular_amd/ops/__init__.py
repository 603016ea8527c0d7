"""Operator namespace mirroring ``nvtabular.ops`` for the hot path
(nvtabular/ops/__init__.py:21-54): the ops named by the north star plus the
graph-plumbing ops the DSL creates."""
from .base import Operator, StatOperator  # noqa: F401
from .bucketize import Bucketize  # noqa: F401
from .categorify import Categorify, get_embedding_sizes  # noqa: F401
from .clip_log import Clip, LogOp  # noqa: F401
from .fill import FillMissing  # noqa: F401
from .groupby import Groupby  # noqa: F401
from .hash_bucket import HashBucket  # noqa: F401
from .hashed_cross import HashedCross  # noqa: F401
from .join_groupby import JoinGroupby  # noqa: F401
from .lambdaop import LambdaOp  # noqa: F401
from .normalize import Normalize, NormalizeMinMax  # noqa: F401
from .selection import ConcatColumns, Rename, SubsetColumns, SubtractionOp  # noqa: F401
from .target_encoding import TargetEncoding  # noqa: F401
from ..selector import ColumnSelector  # noqa: F401
