"""`ouster.sdk.sensor`: live sensors are out of scope; the names exist so that test modules import."""


class ClientError(Exception):
    pass


class ClientTimeout(ClientError):
    pass


class SensorPacketSource:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("live sensor sources are out of scope")
