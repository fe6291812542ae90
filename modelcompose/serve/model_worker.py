"""modelcompose.serve.model_worker: the model-facing half (generate_stream / get_status) on the continuous-batching engine."""
from modelcompose_amd.serve.model_worker import ContinuousBatcher, GenerationRequest, ModelWorker  # noqa: F401
