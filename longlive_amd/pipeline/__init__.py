from .causal_inference import CausalInferencePipeline
from .interactive_causal_inference import InteractiveCausalInferencePipeline
from .streaming_training import StreamingSwitchTrainingPipeline, StreamingTrainingPipeline
from .throughput import InterleavedStreams

__all__ = ["InterleavedStreams", "CausalInferencePipeline", "InteractiveCausalInferencePipeline", "StreamingTrainingPipeline",
           "StreamingSwitchTrainingPipeline"]
