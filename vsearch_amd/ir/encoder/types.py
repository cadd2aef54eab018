"""Encoder registry (/root/reference/src/ir/encoder/types.py:8-21). Only the VDR text tower is on the
vocabulary-space retrieval hot path; DPR and the cross-modal towers are out of scope."""
from .vdr import VDREncoder, VDREncoderConfig

ENCODER_TYPES = {"vdr": VDREncoder}
CONFIG_TYPES = {"vdr": VDREncoderConfig}
