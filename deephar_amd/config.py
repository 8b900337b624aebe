"""Hyper-parameter bags accepted by the builders (reference deephar/config.py).

ModelConfig keeps the reference's attribute names verbatim because spnet.build(cfg) reads and mutates them
(config.py:150-192, spnet.py:376-377).  The data-augmentation half of the reference's config (DataConfig and
its per-dataset instances, config.py:6-148) only matters here for `input_shape`; a minimal stand-in is kept
so callers that pass `mpii_sp_dataconf.input_shape` keep working."""


class ModelConfig(object):
    """Hyperparameters for models (config.py:150-192)."""

    def __init__(self, input_shape, poselayout,
                 num_actions=[],
                 num_pyramids=8,
                 action_pyramids=[1, 2],
                 num_levels=4,
                 kernel_size=(5, 5),
                 growth=96,
                 image_div=8,
                 predict_rootz=False,
                 downsampling_type='maxpooling',
                 pose_replica=False,
                 num_pose_features=128,
                 num_visual_features=128,
                 sam_alpha=1,
                 dbg_decoupled_pose=False,
                 dbg_decoupled_h=False):
        assert type(num_actions) == list, 'num_actions should be a list'
        self.input_shape = input_shape
        self.num_joints = poselayout.num_joints
        self.dim = poselayout.dim
        self.num_actions = num_actions
        self.num_pyramids = num_pyramids
        self.action_pyramids = action_pyramids
        self.num_levels = num_levels
        self.kernel_size = kernel_size
        self.growth = growth
        self.image_div = image_div
        self.predict_rootz = predict_rootz
        self.downsampling_type = downsampling_type
        self.pose_replica = pose_replica
        self.num_pose_features = num_pose_features
        self.num_visual_features = num_visual_features
        self.sam_alpha = sam_alpha
        self.dbg_decoupled_pose = dbg_decoupled_pose
        self.dbg_decoupled_h = dbg_decoupled_h


class DataConfig(object):
    """Only the crop geometry of the reference's DataConfig (config.py:9-40); augmentation is out of scope."""

    def __init__(self, crop_resolution=(256, 256), image_channels=(3,)):
        self.crop_resolution = crop_resolution
        self.image_channels = image_channels
        self.input_shape = crop_resolution + image_channels


mpii_sp_dataconf = DataConfig()
mpii_dataconf = mpii_sp_dataconf
pennaction_dataconf = DataConfig()
pennaction_pe_dataconf = DataConfig()
human36m_dataconf = DataConfig()
ntu_dataconf = DataConfig()
ntu_pe_dataconf = DataConfig()
