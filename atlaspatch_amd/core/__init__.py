from .config import (AppConfig, ExtractionConfig, FeatureExtractionConfig, OutputConfig,
                     ProcessingConfig, SegmentationConfig, VisualizationConfig)
from .models import ExtractionResult, Mask, Slide

__all__ = ["AppConfig", "ExtractionConfig", "FeatureExtractionConfig", "OutputConfig",
           "ProcessingConfig", "SegmentationConfig", "VisualizationConfig",
           "ExtractionResult", "Mask", "Slide"]
