"""The preset-facing parameter objects of rl_coach/base_parameters.py that are not tied to a framework
graph: `Parameters` (:149-178, attributes can only be created in a constructor), the `Frameworks` enum
(:32-34) with the member this engine adds, the embedder / middleware scheme enums (:37-49), and the
`VisualizationParameters` / `PresetValidationParameters` / `TaskParameters` records every preset hands to
its graph manager (:222-305, 436-477, 570-609).  The device engine reads only `dump_csv` and the
validation thresholds; dashboards, video dumps and rendering are out of scope and the flags are inert."""
import json
from enum import Enum


class Frameworks(Enum):
    tensorflow = "TensorFlow"
    mxnet = "MXNet"
    hip = "HIP"                      # this engine: hand-written gfx950 kernels behind the same agents


class EmbedderScheme(Enum):
    Empty = "Empty"
    Shallow = "Shallow"
    Medium = "Medium"
    Deep = "Deep"


class MiddlewareScheme(Enum):
    Empty = "Empty"
    Shallow = "Shallow"
    Medium = "Medium"
    Deep = "Deep"


class DistributedCoachSynchronizationType(Enum):
    SYNC = "sync"
    ASYNC = "async"


class Parameters(object):
    def __setattr__(self, key, value):
        import sys
        if sys._getframe(1).f_code.co_name != '__init__' and not hasattr(self, key):
            raise TypeError("Parameter '{}' does not exist in {}. Parameters are only to be defined in a constructor of"
                            " a class inheriting from Parameters. In order to explicitly register a new parameter "
                            "outside of a constructor use register_var().".format(key, self.__class__))
        object.__setattr__(self, key, value)

    def register_var(self, key, value):
        if hasattr(self, key):
            raise TypeError("Cannot register an already existing parameter '{}'. ".format(key))
        object.__setattr__(self, key, value)

    def __str__(self):
        return "\"{}\" {}\n".format(self.__class__.__name__, json.dumps(self.__dict__, indent=4, default=repr))


class VisualizationParameters(Parameters):
    def __init__(self, print_networks_summary=False, dump_csv=True, dump_signals_to_csv_every_x_episodes=5,
                 dump_gifs=False, dump_mp4=False, video_dump_methods=None, dump_in_episode_signals=False,
                 dump_parameters_documentation=True, render=False, native_rendering=False,
                 max_fps_for_human_control=10, tensorboard=False, add_rendered_image_to_env_response=False):
        self.print_networks_summary = print_networks_summary
        self.dump_csv = dump_csv
        self.dump_signals_to_csv_every_x_episodes = dump_signals_to_csv_every_x_episodes
        self.dump_gifs, self.dump_mp4 = dump_gifs, dump_mp4
        self.video_dump_methods = video_dump_methods or []
        self.dump_in_episode_signals = dump_in_episode_signals
        self.dump_parameters_documentation = dump_parameters_documentation
        self.render, self.native_rendering = render, native_rendering
        self.max_fps_for_human_control = max_fps_for_human_control
        self.tensorboard = tensorboard
        self.add_rendered_image_to_env_response = add_rendered_image_to_env_response


class PresetValidationParameters(Parameters):
    def __init__(self, test=False, min_reward_threshold=0, max_episodes_to_achieve_reward=1, num_workers=1,
                 reward_test_level=None, test_using_a_trace_test=True, trace_test_levels=None,
                 trace_max_env_steps=5000, read_csv_tries=200):
        self.test = test
        self.min_reward_threshold = min_reward_threshold
        self.max_episodes_to_achieve_reward = max_episodes_to_achieve_reward
        self.num_workers = num_workers
        self.reward_test_level = reward_test_level
        self.test_using_a_trace_test = test_using_a_trace_test
        self.trace_test_levels = trace_test_levels
        self.trace_max_env_steps = trace_max_env_steps
        self.read_csv_tries = read_csv_tries


class TaskParameters(Parameters):
    def __init__(self, framework_type=Frameworks.hip, evaluate_only=None, use_cpu=False, experiment_path='/tmp',
                 seed=None, checkpoint_save_secs=None, checkpoint_restore_dir=None, checkpoint_restore_path=None,
                 checkpoint_save_dir=None, export_onnx_graph=False, apply_stop_condition=False, num_gpu=1):
        if use_cpu:
            raise ValueError("the HIP engine has no CPU path (use_cpu=True)")
        self.framework_type = framework_type
        self.task_index = 0
        self.evaluate_only = evaluate_only
        self.use_cpu = use_cpu
        self.experiment_path = experiment_path
        self.checkpoint_save_secs = checkpoint_save_secs
        self.checkpoint_restore_path = checkpoint_restore_path or checkpoint_restore_dir
        self.checkpoint_save_dir = checkpoint_save_dir
        self.seed = seed
        self.export_onnx_graph = export_onnx_graph
        self.apply_stop_condition = apply_stop_condition
        self.num_gpu = num_gpu
