"""GPU counterparts of the reference's scoring stage: score/prob_inference.py (per-frame
view-mean probabilities) and score/sv_level/LiDAL.py (inter-frame divergence / entropy per
supervoxel + greedy selection), with frames sharded over the GPUs of a node."""
from .interframe import FrameBank, neighbour_ids, score_frame
from .pipeline import ScoreBoard, collect_sequence, score_sequence
from .prob_inference import infer_frame
from .selection import select
from .sharding import HaloExchange, frame_range, gather_frames, needed_frames

__all__ = ['infer_frame', 'FrameBank', 'neighbour_ids', 'score_frame', 'score_sequence', 'collect_sequence', 'ScoreBoard', 'select',
           'frame_range', 'gather_frames', 'needed_frames', 'HaloExchange']
