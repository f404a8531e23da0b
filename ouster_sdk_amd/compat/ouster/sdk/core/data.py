"""`ouster.sdk.core.data`: the names the reference defines in Python (python/src/ouster/sdk/core/data.py:69-81)."""
from enum import Enum


class ColHeader(Enum):
    """Column headers available in lidar data (selector of PacketFormat.packet_header)."""
    TIMESTAMP = 0
    ENCODER_COUNT = 1
    MEASUREMENT_ID = 2
    STATUS = 3
    FRAME_ID = 4

    def __int__(self) -> int:
        return self.value
