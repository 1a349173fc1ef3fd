from .sampler import PointwiseSampler
from .sampler import PairwiseSampler
from .dataset import Dataset
