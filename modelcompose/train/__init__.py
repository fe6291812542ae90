from modelcompose_amd.train import MultimodalTrainStep  # noqa: F401
