"""pose-prediction mode (reference: ramp/pose_prediction/)"""
