from .sampler import PointwiseSampler
from .sampler import PairwiseSampler
from .dataset import Dataset
from .sampler import TimeOrderPointwiseSampler
from .sampler import TimeOrderPairwiseSampler
