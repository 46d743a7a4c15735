"""message_pdu: string messages as PDUs (python/message_pdu.py), the dependency of
spectrum_sensor_v2 / multichannel_scanner for their ``freq_msg_PDU`` port
(spectrum_sensor_v2.py:112,182)."""
from .gr_compat import sync_block, to_msg


class message_pdu(sync_block):
    def __init__(self, period=None):
        sync_block.__init__(self, 'message_pdu', [], [])
        self.message_port_register_out('out')

    def post_message(self, key, value):
        self.message_port_pub('out', to_msg(key, value))
