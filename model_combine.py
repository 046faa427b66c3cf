"""Plug-in module for ``--model model_combine`` (main.py:65-66, 108): exposes `Seq2SeqAttNN` backed by the
MI355X HIP engine.  See session-based-news-recommendation_amd/host/model.py."""
import tcar_amd  # noqa: F401
from tcar_amd.host.model import Seq2SeqAttNN  # noqa: F401
