"""Eval driver of the composed model (mirror of modelcompose/eval/model_multimodal_qa_loader.py)."""
