from .causal_inference import CausalInferencePipeline
from .interactive_causal_inference import InteractiveCausalInferencePipeline
from .streaming_training import StreamingSwitchTrainingPipeline, StreamingTrainingPipeline

__all__ = ["CausalInferencePipeline", "InteractiveCausalInferencePipeline", "StreamingTrainingPipeline",
           "StreamingSwitchTrainingPipeline"]
