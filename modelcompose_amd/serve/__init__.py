from .model_worker import ContinuousBatcher, GenerationRequest, ModelWorker  # noqa: F401
