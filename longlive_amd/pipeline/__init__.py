from .causal_inference import CausalInferencePipeline
from .interactive_causal_inference import InteractiveCausalInferencePipeline

__all__ = ["CausalInferencePipeline", "InteractiveCausalInferencePipeline"]
