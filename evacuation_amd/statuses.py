"""Status codes (reference: src/env/env/statuses.py:16-27, Enum auto() values)."""
from enum import Enum


class Status(Enum):
    VISCEK = 1      # pedestrian under Vicsek rules
    FOLLOWER = 2    # follower of the leader (agent)
    EXITING = 3     # pedestrian in the exit zone
    ESCAPED = 4     # evacuated pedestrian

    @classmethod
    def all(cls):
        return list(cls)


# reference: src/env/constants.py:35-38
SWITCH_DISTANCE_TO_LEADER = 0.2
SWITCH_DISTANCE_TO_OTHER_PEDESTRIAN = 0.1
SWITCH_DISTANCE_TO_EXIT = 0.4
SWITCH_DISTANCE_TO_ESCAPE = 0.01
