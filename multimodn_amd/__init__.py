"""multimodn_amd: MI355X-native implementation of MultiModN's sequential-fusion training hot
path behind the reference's plugin surface (MultiModN / MultiModEncoder / MultiModDecoder /
InitState / MultiModDataset / MultiModNHistory)."""
from .state import InitState, TrainableInitState, StaticInitState
from .encoders import MultiModEncoder, MLPEncoder, MLPFeatureEncoder, MIMIC_MLPEncoder, SLPEncoder, LinearEncoder, LogisticEncoder
from .decoders import MultiModDecoder, ClassDecoder, MLPDecoder, LogisticDecoder
from .history import MultiModNHistory
from .datasets import MultiModDataset, PartitionDataset, FeatureWiseDataset, JointDatasets, DeviceResidentLoader
from .multimodn import MultiModN
from .engine import HipChainEngine, UnsupportedModelError
from . import optim, metrics

__all__ = [
    "InitState", "TrainableInitState", "StaticInitState", "MultiModEncoder", "MLPEncoder", "SLPEncoder",
    "LinearEncoder", "LogisticEncoder", "MultiModDecoder", "ClassDecoder", "LogisticDecoder", "MIMIC_MLPEncoder",
    "MLPDecoder", "MLPFeatureEncoder",
    "MultiModNHistory", "MultiModDataset", "PartitionDataset", "FeatureWiseDataset", "JointDatasets",
    "MultiModN", "HipChainEngine", "UnsupportedModelError", "optim", "DeviceResidentLoader", "metrics",
]
